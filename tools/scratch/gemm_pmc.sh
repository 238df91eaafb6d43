out=$GRAFT_REPO_ROOT/gpurun_out/r2v/wg; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for what in ${1:-wgrad}; do
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $out/p1 -o pmc --output-format csv -- python $GRAFT_REPO_ROOT/tools/gemm_pmc_probe.py $what 322624 > $out/p1.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS -d $out/p2 -o pmc --output-format csv -- python $GRAFT_REPO_ROOT/tools/gemm_pmc_probe.py $what 322624 > $out/p2.log 2>&1
done
cd $GRAFT_REPO_ROOT
python - <<'P'
import csv,glob,collections
for d in ('p1','p2'):
    for f in glob.glob('gpurun_out/r2v/wg/%s/**/*counter_collection.csv'%d, recursive=True):
        rows=list(csv.DictReader(open(f)))
        agg=collections.defaultdict(lambda: collections.defaultdict(list))
        for r in rows:
            agg[r['Kernel_Name'][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
        for k,v in agg.items():
            if 'wgrad' in k or 'gemm' in k:
                print(k, {c: sum(x)/len(x) for c,x in v.items()}, len(next(iter(v.values()))))
P
