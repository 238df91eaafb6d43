#!/bin/bash
# rocprofv3 kernel trace + stats of the SpMM microbenchmarks (config 5 packed / padded, config 4) for profiles/.
out=${1:-gpurun_out/prof_mb}; tag=${2:-r2}
mkdir -p $out
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for wl in c5 c4; do
  rocprofv3 --kernel-trace --stats -d $root/$out/$wl -o mb --output-format csv -- python $root/tools/spmm_microbench.py $wl > $root/$out/${tag}_spmm_microbench_$wl.txt 2>&1
  cp $(find $root/$out/$wl -name "*kernel_stats.csv") $root/$out/${tag}_spmm_microbench_${wl}_kernel_stats.csv
done
SN_MB_LAYOUT=padded rocprofv3 --kernel-trace --stats -d $root/$out/c5pad -o mb --output-format csv -- python $root/tools/spmm_microbench.py c5 > $root/$out/${tag}_spmm_microbench_c5_padded.txt 2>&1
cp $(find $root/$out/c5pad -name "*kernel_stats.csv") $root/$out/${tag}_spmm_microbench_c5_padded_kernel_stats.csv
