#!/bin/bash
# kernel stats of an arbitrary python command line.  usage: trace_script.sh OUTDIR script.py [args]
out=$1; shift; mkdir -p $out; root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $root/$out/trace -o tb --output-format csv -- python $root/$@ > $root/$out/trace.log 2>&1
cd $root; cp $(find $out/trace -name "*kernel_stats.csv") $out/kernel_stats.csv; rm -rf $out/trace
