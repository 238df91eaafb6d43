#!/bin/bash
# Profiles of the default bench step for profiles/: kernel trace + stats, then in-step PMC traffic (separate passes).
# usage: tools/profile_bench.sh OUTDIR TAG        (run on the GPU box through gpurun)
out=${1:-gpurun_out/prof_bench}; tag=${2:-r2}
mkdir -p $out
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
args="--steps 6 --warmup 2 --no-cpu-baseline --no-secondary --no-pmc"
rocprofv3 --kernel-trace --stats -d $root/$out/trace -o bench --output-format csv -- python $root/bench.py $args > $root/$out/trace.log 2>&1
rocprofv3 --pmc TCC_EA0_RDREQ_sum -d $root/$out/rd -o pmc --output-format csv -- python $root/bench.py $args > $root/$out/rd.log 2>&1
rocprofv3 --pmc TCC_EA0_WRREQ_sum -d $root/$out/wr -o pmc --output-format csv -- python $root/bench.py $args > $root/$out/wr.log 2>&1
cd $root
python tools/pmc_bench.py $out/${tag}_pmc_traffic_c3 $(find $out/rd -name "*counter_collection.csv") $(find $out/wr -name "*counter_collection.csv") last=6 > $out/pmc_summary.txt 2>&1
cp $(find $out/trace -name "*kernel_stats.csv") $out/${tag}_bench_c3_kernel_stats.csv
grep "^{" $out/trace.log > $out/${tag}_bench_c3_under_rocprof.json
