#!/bin/bash
# PMC passes over tools/lap_probe.py, one counter group per pass (never combined with tracing).  usage: lap_pmc.sh OUTDIR
out=${1:-gpurun_out/lap_pmc}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_TAG_STALL_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" \
           "TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum" "SQ_WAVES SQ_BUSY_CYCLES" \
           "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" \
           "TCP_TCC_READ_REQ_LATENCY_sum GRBM_GUI_ACTIVE" "TCC_BUSY_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $grp -d $GRAFT_REPO_ROOT/$out/p$i -o pmc --output-format csv -- python $GRAFT_REPO_ROOT/tools/lap_probe.py > $GRAFT_REPO_ROOT/$out/p$i.log 2>&1
done
cd $GRAFT_REPO_ROOT
python tools/summarize_pmc.py $out/summary.csv 10 $(find $out -name "*counter_collection.csv" | sort)
