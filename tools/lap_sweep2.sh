#!/bin/bash
# RB4 kernel sweep: gathers in flight per lane (KB) x passes per wave.
out=${1:-gpurun_out/lap_sweep2.txt}
: > $out
for wl in c4 c5; do
  for kb in 4 8 12 16; do
    for it in 1 2 4; do
      echo "== $wl rb4 KB=$kb iters=$it" >> $out
      SN_MB_ONLY=L SN_RB4_KB=$kb SN_RB4_ITERS=$it python tools/spmm_microbench.py $wl 2>&1 | grep " rb4 " >> $out
    done
  done
done
