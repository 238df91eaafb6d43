#!/bin/bash
# kernel trace of the default bench step + the launch sequence of one steady-state step (tools/trace_steps.py)
out=${1:-gpurun_out/trace_seq}; mkdir -p $out; root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $root/$out/trace -o bench --output-format csv -- python $root/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary > $root/$out/trace.log 2>&1
cd $root
python tools/trace_steps.py $(find $out/trace -name "*kernel_trace.csv") $out/step_seq.txt > $out/steps.txt 2>&1
rm -rf $out/trace
