#!/bin/bash
# A/B of the default bench step on ONE box: tools/scratch/ab_bench.sh OUT "ENV_A" "ENV_B" [reps]   (ENV_x: e.g. "SN_WGRAD_H=-1")
out=$1; a=$2; b=$3; reps=${4:-2}; mkdir -p $out
for r in $(seq $reps); do
  for tag in A B; do
    if [ $tag = A ]; then e="$a"; else e="$b"; fi
    env $e python bench.py --no-cpu-baseline --no-secondary > $out/bench_$tag$r.json 2> $out/bench_$tag$r.err
    python - <<PY
import json
d = json.loads(open("$out/bench_$tag$r.json").read().strip().splitlines()[-1])
print("$tag$r [$e]", round(d["value"], 1), "meshes/s", round(d["ms_per_step"], 3), "ms/step  linear", round(d["roofline"]["linear_ms_per_step"], 3), " spmm", round(d["roofline"]["spmm_ms_per_step_all_kernels"], 3))
PY
  done
done
