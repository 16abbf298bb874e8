#!/bin/bash
# Same-box A/B of this tree against another tree of the repository checked out (and built) under _prev/
# (git worktree add _prev <commit>; cd _prev && python -c "import __graft_entry__ as g; g.build()"): bench.py of each, alternating.
#   scripts/ab_prev_tree.sh [configs, default "2 3 5"]
run() {  # tree config steps warmup windows
  d=.; [ $1 = prev ] && d=_prev
  (cd $d && python bench.py --config $2 --steps $3 --warmup $4 --windows $5 --no-also --no-cpu-baseline 2>/dev/null | tail -1 |
     python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 c$2', round(d['ms_per_step'],5), 'min', round(d['timing']['ms_per_step_min'],5))")
}
for c in ${1:-2 3 5}; do
  s=50; w=10; n=5; [ $c = 3 ] && s=20; [ $c = 5 ] && { s=3; w=1; n=2; }
  for i in 1 2 3; do
    if [ $((i % 2)) = 1 ]; then run prev $c $s $w $n; run cur $c $s $w $n; else run cur $c $s $w $n; run prev $c $s $w $n; fi
  done
done
