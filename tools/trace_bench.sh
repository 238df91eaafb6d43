#!/bin/bash
# Kernel trace + stats of the default bench step only (no PMC passes).  usage: tools/trace_bench.sh OUTDIR [extra bench args]
out=${1:-gpurun_out/trace_bench}; shift
mkdir -p $out
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $root/$out/trace -o bench --output-format csv -- python $root/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary "$@" > $root/$out/trace.log 2>&1
cd $root
cp $(find $out/trace -name "*kernel_stats.csv") $out/kernel_stats.csv
rm -rf $out/trace
