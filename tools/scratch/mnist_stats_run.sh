root=$GRAFT_REPO_ROOT; out=gpurun_out/mn64; mkdir -p $root/$out; cd $root
for n in 10 60; do (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $root/$out/tr_$n -o t -- python $root/tools/scratch/mnist_replay_only.py $n > $root/$out/tr_$n.log 2>&1); done
python tools/scratch/replay_stats.py $(find $out/tr_10 -name "*kernel_stats.csv") 10 $(find $out/tr_60 -name "*kernel_stats.csv") 60 $out/mnist64.csv
rm -rf $out/tr_10 $out/tr_60
