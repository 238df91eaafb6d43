#!/bin/bash
# SQ counters of the forward Linear kernels (tools/gemm_pmc_probe.py), one pass per counter group; prints per-kernel means.
root=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
for mode in fwd_copy fwd_elu dgrad; do
  for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM_WR" \
             "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD"; do
    rm -rf /tmp/gp; rocprofv3 --pmc $grp -d /tmp/gp -o p --output-format csv -- python $root/tools/gemm_pmc_probe.py $mode > /dev/null 2>&1
    f=$(find /tmp/gp -name '*counter_collection.csv' | head -1)
    python - "$f" "$mode" <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "gemm_" in r["Kernel_Name"]]
agg = collections.defaultdict(list)
for r in rows: agg[(r["Kernel_Name"].split("(")[0][-60:], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(agg.items()): print(f"{sys.argv[2]:9s} {k:60s} {c:28s} {sum(v)/len(v):16.0f}  (n={len(v)})")
PY
  done
done
