#!/bin/bash
root=$GRAFT_REPO_ROOT
cd $root/tools/scratch/gapbench && hipcc --offload-arch=gfx950 -O2 -o /tmp/gap gap.hip || exit 1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/gaptrace
rocprofv3 --kernel-trace -d /tmp/gaptrace -o gap --output-format csv -- /tmp/gap > $root/gpurun_out/gap_wall.txt 2>&1
python $root/tools/scratch/gapbench/gaps.py $(find /tmp/gaptrace -name "*kernel_trace.csv") > $root/gpurun_out/gap_pairs.txt 2>&1
/tmp/gap > $root/gpurun_out/gap_wall_untraced.txt 2>&1
