out=$GRAFT_REPO_ROOT/gpurun_out/r2v/wg2; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
what=${1:-wgrad}
for wv in 3 2 1; do
export SN_WGRAD_VARIANT=$wv
rocprofv3 --pmc SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $out/a$wv -o pmc --output-format csv -- python $GRAFT_REPO_ROOT/tools/gemm_pmc_probe.py $what 322624 > $out/a$wv.log 2>&1
rocprofv3 --pmc TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_SERIALIZATION_STALL_sum TCP_TA_TCP_STATE_READ_sum GRBM_GUI_ACTIVE -d $out/b$wv -o pmc --output-format csv -- python $GRAFT_REPO_ROOT/tools/gemm_pmc_probe.py $what 322624 > $out/b$wv.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES -d $out/c$wv -o pmc --output-format csv -- python $GRAFT_REPO_ROOT/tools/gemm_pmc_probe.py $what 322624 > $out/c$wv.log 2>&1
done
cd $GRAFT_REPO_ROOT
python - <<'P'
import csv,glob,collections
for wv in (3,2,1):
  for d in ('a','b','c'):
    for f in glob.glob('gpurun_out/r2v/wg2/%s%d/**/*counter_collection.csv'%(d,wv), recursive=True):
        rows=list(csv.DictReader(open(f)))
        agg=collections.defaultdict(lambda: collections.defaultdict(list))
        for r in rows:
            agg[r['Kernel_Name'][:40]][r['Counter_Name']].append(float(r['Counter_Value']))
        for k,v in agg.items():
            if 'wgrad_' in k and 'reduce' not in k:
                print(wv, d, k[28:], {c: round(sum(x)/len(x)) for c,x in v.items()})
P
