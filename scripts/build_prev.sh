#!/bin/bash
# builds HEAD's sources into opfgym_amd/libopfx_prev.so (A/B partner for scripts/ab.sh)
set -e
rm -rf /tmp/prev; git worktree add -f /tmp/prev ${1:-HEAD} -q
(cd /tmp/prev && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -mllvm -disable-machine-licm -Iinclude -Iopfgym_amd/csrc -o /root/repo/opfgym_amd/libopfx_prev.so opfgym_amd/csrc/plan.cpp opfgym_amd/csrc/opfx.hip)
git worktree remove --force /tmp/prev
echo built prev
