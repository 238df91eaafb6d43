#!/bin/bash
# in-step A/B/C... on ONE box: tools/scratch/ab3_bench.sh OUT reps "ENV1" "ENV2" ...
out=$1; reps=$2; shift 2; mkdir -p $out
for r in $(seq $reps); do
  i=0
  for e in "$@"; do
    i=$((i+1))
    env $e python bench.py --no-cpu-baseline --no-secondary > $out/bench_${i}_$r.json 2> $out/bench_${i}_$r.err
    python - <<PY
import json
d = json.loads(open("$out/bench_${i}_$r.json").read().strip().splitlines()[-1])
print("cfg$i r$r [$e]", round(d["value"], 1), "meshes/s", round(d["ms_per_step"], 3), "ms/step  linear", round(d["roofline"]["linear_ms_per_step"], 3), " spmm", round(d["roofline"]["spmm_ms_per_step_all_kernels"], 3))
PY
  done
done
