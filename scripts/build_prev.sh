#!/bin/bash
# builds a commit's sources (default HEAD) into opfgym_amd/libopfx_prev.so (A/B partner for scripts/ab.sh)
set -e
rm -rf /tmp/prev; git worktree add -f /tmp/prev ${1:-HEAD} -q
(cd /tmp/prev && python -c "import __graft_entry__ as g; g._compile(g.SOURCES, g.OUT, force=True)" && cp opfgym_amd/libopfx.so /root/repo/opfgym_amd/libopfx_prev.so)
git worktree remove --force /tmp/prev
echo built prev
