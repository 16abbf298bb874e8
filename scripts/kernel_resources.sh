#!/bin/bash
# Registers / scratch / LDS of every kernel as the compiler reports them (no GPU needed).  Arguments: kernel translation units
# of opfgym_amd/csrc (default: all k_*.hip), then any extra compiler flags after `--`.
cd "$(dirname "$0")/.."
units=(); flags=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do units+=("$1"); shift; done
[ "$1" == "--" ] && shift; flags=("$@")
[ ${#units[@]} -eq 0 ] && units=(opfgym_amd/csrc/k_*.hip)
for u in "${units[@]}"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -disable-machine-licm -Iinclude -Iopfgym_amd/csrc \
    "${flags[@]}" -c "$u" -o /tmp/opfx_res.o -Rpass-analysis=kernel-resource-usage 2>&1 |
  awk '/Function Name/ {n=$0; sub(/.*Function Name: /,"",n); sub(/ \[.*/,"",n)}
       /VGPRs:/ && !/AGPRs|Spill/ {v=$NF} / VGPRs: / {v=$(NF-1)}
       /AGPRs:/ {a=$(NF-1)} /ScratchSize/ {s=$(NF-1)} /Occupancy/ {o=$(NF-1)}
       /LDS Size/ {print n, "VGPR", v, "AGPR", a, "scratch", s, "occupancy", o}' | c++filt | sed 's/(anonymous namespace):://g; s/(.*)//'
done
