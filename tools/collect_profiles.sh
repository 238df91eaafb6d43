#!/bin/bash
# Evidence of one round, collected on the GPU box into gpurun_out/$1/ (copy what is to be judged into profiles/):
#   tools/collect_profiles.sh r5_final [tests|bench|trace|pmc|configs ...]      (default: everything)
# rocprofv3 runs from /tmp with TMPDIR=/tmp; counter passes (--pmc) are separate runs without any trace domain.
[ -e /dev/kfd ] || { echo "no GPU here: run this through gpurun (gpurun -- bash tools/collect_profiles.sh ...)" >&2; exit 2; }
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
tag=${1:-r5_final}; shift
what=${*:-tests bench trace pmc q3pmc plans configs}
out=$root/gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
kstats() {   # kstats NAME -- command...: rocprofv3 kernel statistics of the command -> $out/NAME_kernel_stats.csv
  name=$1; shift; shift
  rm -rf /tmp/ks_$name
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$name -o t -- "$@" > $out/${name}_under_rocprof.log 2>&1
  f=$(find /tmp/ks_$name -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $out/${name}_kernel_stats.csv
  rm -rf /tmp/ks_$name
}
for w in $what; do case $w in
tests)
  (cd $root && timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -6) > $out/gpu_tests.txt ;;
bench)
  (cd $root && timeout 1500 python bench.py > $out/bench_c3_final.json 2> $out/bench_c3_final.err; cp bench_detail.json $out/bench_c3_final_detail.json)
  (cd $root && timeout 600 python bench.py --graph --no-cpu-baseline --no-secondary --no-pmc > $out/bench_c3_graph.json 2>/dev/null)
  (cd $root && timeout 600 python bench.py --workload faust > $out/bench_faust_n1.json 2>/dev/null)
  (cd $root && timeout 900 python bench.py --gpus 2 --steps 10 --warmup 3 --no-secondary --no-cpu-baseline > $out/bench_gpus2_gloo_one_device.json 2>/dev/null) ;;
trace)
  kstats bench_c3 -- python $root/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-secondary --no-pmc
  grep '^{' $out/bench_c3_under_rocprof.log > $out/bench_c3_under_rocprof.json; rm -f $out/bench_c3_under_rocprof.log
  for o in grid permuted permuted_both permuted_both+reorder; do
    n=spmm_microbench_c5_$(echo $o | tr '+' '_')
    kstats $n -- python $root/tools/order_probe.py c5 order=$o
    grep -v amdgpu.ids $out/${n}_under_rocprof.log > $out/$n.json; rm -f $out/${n}_under_rocprof.log
  done ;;
orders)
  for o in grid permuted permuted_both permuted_both+reorder; do
    n=spmm_microbench_c5_$(echo $o | tr '+' '_')
    kstats $n -- python $root/tools/order_probe.py c5 order=$o
    grep -v amdgpu.ids $out/${n}_under_rocprof.log > $out/$n.json; rm -f $out/${n}_under_rocprof.log
  done ;;
pmc)
  for ctr in TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum; do
    rm -rf /tmp/pmc_$ctr
    rocprofv3 --pmc $ctr -d /tmp/pmc_$ctr -o pmc --output-format csv -- python $root/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary --no-pmc --backend none > /dev/null 2>&1
  done
  rd=$(find /tmp/pmc_TCC_EA0_RDREQ_sum -name '*counter_collection.csv' | head -1); wr=$(find /tmp/pmc_TCC_EA0_WRREQ_sum -name '*counter_collection.csv' | head -1)
  (cd $root && python tools/pmc_bench.py $out/pmc_traffic_c3 $rd $wr last=6 > $out/pmc_traffic_c3_top.txt 2>&1)
  rm -rf /tmp/pmc_TCC_EA0_RDREQ_sum /tmp/pmc_TCC_EA0_WRREQ_sum ;;
q3pmc)
  # SQ / TCC counters of the four config-5 Dirac products, one counter group per pass (never with a trace domain): what the
  # face-output (write-heavy) shapes wait for — LABNOTES "write-heavy Dirac shapes"
  i=0
  for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
             "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_WRREQ_STALL_sum TCC_TAG_STALL_sum" \
             "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum" "TCP_PENDING_STALL_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
             "TCC_BUSY_sum GRBM_GUI_ACTIVE"; do
    i=$((i+1)); rm -rf /tmp/q3pmc_$i
    rocprofv3 --pmc $grp -d /tmp/q3pmc_$i -o pmc --output-format csv -- python $root/tools/q3_counters.py 6 > $out/q3pmc_$i.log 2>&1
  done
  (cd $root && python tools/summarize_pmc.py $out/q3_counters_c5.csv 6 $(find /tmp/q3pmc_* -name '*counter_collection.csv' | sort) > $out/q3_counters_c5.txt 2>&1)
  grep -h '^{' $out/q3pmc_1.log > $out/q3_counters_c5_plan.json; rm -rf /tmp/q3pmc_* ;;
plans)
  (cd $root && timeout 900 python tools/plan_probe.py arap arap4 mnist faust 2>/dev/null | tail -1) > $out/plan_probe.json
  (cd $root && timeout 900 python tools/plan_probe.py faust mnist arap4 --sections 2>/dev/null | grep -v amdgpu.ids) > $out/plan_probe_sections.txt ;;
configs)
  (cd $root && for c in mnist_dir mnist_lap faust_lap arap_lap arap_ragged arap_swap mnist_swap faust_swap; do timeout 600 python tools/train_bench.py $c 20 2>/dev/null | grep -v amdgpu.ids; done
   SN_SWAP_MODEL=product timeout 600 python tools/train_bench.py arap_swap 20 2>/dev/null | grep -v amdgpu.ids
   SN_RESIDENT=0 SN_SWAP_MESHES=8 timeout 600 python tools/train_bench.py arap_swap 2 2>/dev/null | grep -v amdgpu.ids
   timeout 300 python tools/small_gemm_probe.py 2>/dev/null | grep -v amdgpu.ids) > $out/train_bench_other_configs.txt ;;
esac; done
ls -la $out
