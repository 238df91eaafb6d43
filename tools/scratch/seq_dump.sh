#!/bin/bash
# Ordered kernel sequence (start offset, duration, name) of one timed step of the default bench.  usage: seq_dump.sh OUT.txt [bench args]
out=${1:-gpurun_out/seq_now.txt}; shift
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/seqtrace
rocprofv3 --kernel-trace -d /tmp/seqtrace -o bench --output-format csv -- python $root/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary --no-pmc "$@" > $root/gpurun_out/seq_bench.log 2>&1
cd $root
python tools/trace_steps.py $(find /tmp/seqtrace -name "*kernel_trace.csv") $out > ${out%.txt}_summary.txt 2>&1
