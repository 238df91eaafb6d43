#!/bin/bash
# Every profile artefact of a round in one GPU call (run through gpurun from the repo root):
#   tools/round_profiles.sh r4        -> gpurun_out/prof_r4/*  (copy what is to be judged into profiles/)
tag=${1:-r4}
root=$GRAFT_REPO_ROOT
out=gpurun_out/prof_$tag
mkdir -p $root/$out
cd $root
python -m pytest tests -q -m gpu -x 2>&1 | tail -3 > $out/${tag}_gpu_tests.txt
bash tools/profile_bench.sh $out $tag
cp $out/${tag}_pmc_traffic_c3.json profiles/ 2>/dev/null          # (this box's copy: bench.py below reads it)
python bench.py > $out/${tag}_bench_c3_final.json 2> $out/bench_final.err
python bench.py --graph --no-cpu-baseline --no-secondary --no-pmc > $out/${tag}_bench_c3_graph.json 2>> $out/bench_final.err
# BASELINE configs[3] as the N-rank job: one rank (RCCL), and two ranks sharing this box's one device (gloo, flagged)
python bench.py --workload faust > $out/${tag}_bench_faust_n1.json 2>> $out/bench_final.err
python bench.py --workload faust --gpus 2 > $out/${tag}_bench_faust_n2_one_device.json 2>> $out/bench_final.err
python bench.py --gpus 2 --no-cpu-baseline --no-secondary > $out/${tag}_bench_gpus2_gloo_one_device.json 2>> $out/bench_final.err
for cfg in faust mnist; do          # capture once + N replays, N = 10 and 60: the difference is 50 replayed steps alone
  for n in 10 60; do
    (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $root/$out/tr_${cfg}_$n -o t -- python $root/tools/scratch/${cfg}_replay_only.py $n > $root/$out/tr_${cfg}_$n.log 2>&1)
  done
  python tools/scratch/replay_stats.py $(find $out/tr_${cfg}_10 -name "*kernel_stats.csv") 10 $(find $out/tr_${cfg}_60 -name "*kernel_stats.csv") 60 $out/${tag}_replay_${cfg}_kernel_stats.csv >> $out/replay.log 2>&1
  rm -rf $out/tr_${cfg}_10 $out/tr_${cfg}_60
done
{
  for cfg in mnist_dir mnist_lap faust_lap arap_lap arap_ragged arap_swap; do python tools/train_bench.py $cfg 40 2>&1 | grep -v amdgpu.ids; done
  echo "--- A/B switches ---"
  SN_PAIR_FUSED=0 python tools/train_bench.py faust_lap 40 2>&1 | grep replay | sed 's/^/SN_PAIR_FUSED=0 (bmm + sn_pair_ce_*): /'
  SN_TWO_STREAM_TOWERS=0 python tools/train_bench.py faust_lap 40 2>&1 | grep replay | sed 's/^/SN_TWO_STREAM_TOWERS=0: /'
  SN_FOLD_PARTS=1 python tools/train_bench.py faust_lap 40 2>&1 | grep replay | sed 's/^/SN_FOLD_PARTS=1: /'
  SN_FOLD_PARTS=1 python tools/train_bench.py mnist_dir 40 2>&1 | grep replay | sed 's/^/SN_FOLD_PARTS=1: /'
  SN_LAP_FORMAT=rb4 python tools/train_bench.py arap_lap 40 2>&1 | grep -v amdgpu.ids | sed 's/^/SN_LAP_FORMAT=rb4: /'
} > $out/${tag}_train_bench_other_configs.txt 2>&1
{ SN_MB_ONLY=L python tools/spmm_microbench.py c5 2>&1 | grep -v amdgpu.ids; python tools/spmm_microbench.py c4 2>&1 | grep -v amdgpu.ids; } > $out/${tag}_spmm_microbench_laplacian.txt 2>&1
python tools/scratch/pair_fused_time.py > $out/${tag}_pair_fused_vs_materialised.txt 2>&1
ls -la $out
