#!/bin/bash
# kernel stats of a tools/train_bench.py workload.  usage: trace_tool.sh OUTDIR workload
out=$1; w=$2; mkdir -p $out; root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $root/$out/trace -o tb --output-format csv -- python $root/tools/train_bench.py $w > $root/$out/trace.log 2>&1
cd $root; cp $(find $out/trace -name "*kernel_stats.csv") $out/kernel_stats.csv; rm -rf $out/trace
