# kernel statistics of a short bench run: tools/scratch/bench_kstats.sh OUTNAME
root=$GRAFT_REPO_ROOT; out=gpurun_out/$1; mkdir -p $root/$out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $root/$out -o b -- python $root/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-secondary > $root/$out/log.txt 2>&1
rm -f $root/$out/*kernel_trace.csv
