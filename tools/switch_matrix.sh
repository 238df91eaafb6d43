#!/bin/bash
# The GPU suite under every switch of include/sn_spmm.h's SWITCHES line, one line per leg into gpurun_out/$1/switch_matrix.txt
# (run on the GPU box: gpurun -- 'bash tools/switch_matrix.sh r6').  Failing test ids of every leg are listed under its line.
[ -e /dev/kfd ] || { echo "no GPU here: run this through gpurun (gpurun -- bash tools/switch_matrix.sh ...)" >&2; exit 2; }
tag=${1:-scratch}; root=$(cd "$(dirname "$0")/.." && pwd); out=$root/gpurun_out/$tag; mkdir -p $out
cd $root
: > $out/switch_matrix.txt
for leg in "SN_PLANS=0" "SN_PLAN_GRAPHS=0" "SN_DEBUG_VALIDATE=1" "SN_STRICT=1" "SN_RESIDENT=0" "SN_RESIDENT_MAX_GB=0.05" "SN_PAIR_FUSED=0" "SN_GEMM_VARIANT=0" "SN_GEMM_VARIANT=1"; do
  log=/tmp/leg_$(echo $leg | tr '=.' '__').log
  env $leg timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $log 2>&1
  echo "$leg: $(tail -1 $log)" >> $out/switch_matrix.txt
  grep "^FAILED\|^ERROR" $log | cut -c1-200 >> $out/switch_matrix.txt
done
cat $out/switch_matrix.txt | cut -c1-160
