#!/bin/bash
# Registers / scratch / LDS of every kernel of opfx.hip as the compiler reports them (no GPU needed).
cd "$(dirname "$0")/.." && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -disable-machine-licm -Iinclude -Iopfgym_amd/csrc \
  "$@" -c opfgym_amd/csrc/opfx.hip -o /tmp/opfx_res.o -Rpass-analysis=kernel-resource-usage 2>&1 |
  awk '/Function Name/ {n=$0; sub(/.*Function Name: /,"",n); sub(/ \[.*/,"",n)}
       /VGPRs:/ && !/AGPRs|Spill/ {v=$NF} / VGPRs: / {v=$(NF-1)}
       /AGPRs:/ {a=$(NF-1)} /ScratchSize/ {s=$(NF-1)} /Occupancy/ {o=$(NF-1)}
       /LDS Size/ {print n, "VGPR", v, "AGPR", a, "scratch", s, "occupancy", o}' | c++filt | sed 's/(anonymous namespace):://g; s/(.*)//'
