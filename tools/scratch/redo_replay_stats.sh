tag=r3; root=$GRAFT_REPO_ROOT; out=gpurun_out/prof_$tag; mkdir -p $root/$out; cd $root
for cfg in faust mnist; do
  for n in 10 60; do
    (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $root/$out/tr_${cfg}_$n -o t -- python $root/tools/scratch/${cfg}_replay_only.py $n > $root/$out/tr_${cfg}_$n.log 2>&1)
  done
  python tools/scratch/replay_stats.py $(find $out/tr_${cfg}_10 -name "*kernel_stats.csv") 10 $(find $out/tr_${cfg}_60 -name "*kernel_stats.csv") 60 $out/${tag}_replay_${cfg}_kernel_stats.csv
  rm -rf $out/tr_${cfg}_10 $out/tr_${cfg}_60
done
{ SN_MB_ONLY=L python tools/spmm_microbench.py c5 2>&1 | grep -v amdgpu.ids; python tools/spmm_microbench.py c4 2>&1 | grep -v amdgpu.ids; python tools/spmm_microbench.py c3 2>&1 | grep -v amdgpu.ids | grep " L "; } > $out/${tag}_spmm_microbench_laplacian.txt 2>&1
tail -12 $out/${tag}_spmm_microbench_laplacian.txt
