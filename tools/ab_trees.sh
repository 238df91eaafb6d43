#!/bin/bash
# Same-box A/B of the headline step between this tree and an earlier commit (boxes of the pool differ by 3 % on the device and
# 1.5x on the host, so only alternating runs on ONE box say anything about a few tenths of a millisecond):
#   here   : git worktree add -f tools/scratch/abtree <commit> && (cd tools/scratch/abtree && python -c "import __graft_entry__ as g; g.build()")
#   on GPU : gpurun -- 'bash tools/ab_trees.sh tools/scratch/abtree 3'
# Round 6 found a +0.33 ms regression this way (a 321 MB copy of a placeholder per Dirac block) that no test and no profile of the
# new tree alone had shown.
other=${1:-tools/scratch/abtree}; reps=${2:-3}
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],3), 'ms/step', round(d['value'],1), 'meshes/s')"; }
for i in $(seq $reps); do
  (cd $root && python bench.py --no-secondary --no-cpu-baseline --no-pmc 2>/dev/null | line "this tree ")
  (cd $root/$other && python bench.py --no-secondary --no-cpu-baseline --no-pmc 2>/dev/null | line "other tree")
done
