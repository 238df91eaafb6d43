#!/bin/bash
out=${1:-gpurun_out/lap_sweep3.txt}
: > $out
for wl in c4 c5; do
  for occ in 0 2 3 4 6; do
    for kb in 2 4 8; do
      echo "== $wl rb4 occ=$occ KB=$kb iters=1" >> $out
      SN_MB_ONLY=L SN_RB4_OCC=$occ SN_RB4_KB=$kb SN_RB4_ITERS=1 python tools/spmm_microbench.py $wl 2>&1 | grep " rb4 " >> $out
    done
  done
done
