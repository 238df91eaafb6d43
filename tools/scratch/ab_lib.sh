#!/bin/bash
# in-step A/B of two builds of the library on ONE box: ab_lib.sh OUT reps libA.so libB.so   (the last one stays installed)
out=$1; reps=$2; shift 2; mkdir -p $out
for r in $(seq $reps); do
  for lib in "$@"; do
    cp $lib surfacenetworks_amd/libsn_hip.so
    python bench.py --no-cpu-baseline --no-secondary --no-pmc > $out/bench_$(basename $lib .so)_$r.json 2> $out/bench_$(basename $lib .so)_$r.err
    python - <<PY
import json
d = json.loads(open("$out/bench_$(basename $lib .so)_$r.json").read().strip().splitlines()[-1])
print("$(basename $lib .so) r$r", round(d["value"], 1), "meshes/s", round(d["ms_per_step"], 3), "ms/step  linear", round(d["roofline"]["linear_ms_per_step"], 3), " spmm", round(d["roofline"]["spmm_ms_per_step_all_kernels"], 3))
PY
  done
done
