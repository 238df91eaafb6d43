#!/bin/bash
# Laplacian (CSR, N = 128) kernel sweep: previous generation (variant 1) vs the rows kernel at every passes-per-wave setting.
out=${1:-gpurun_out/lap_sweep.txt}
: > $out
for wl in c4 c5; do
  echo "== $wl variant 1 (spmm_csr_lds)" >> $out
  SN_MB_ONLY=L SN_CSR_VARIANT=1 python tools/spmm_microbench.py $wl 2>&1 | grep " L " >> $out
  for it in 0 1 2 4 8 16; do
    echo "== $wl rows / rb4 kernels SN_CSR_ITERS=SN_RB4_ITERS=$it" >> $out
    SN_MB_ONLY=L SN_CSR_VARIANT=2 SN_CSR_ITERS=$it SN_RB4_ITERS=$it python tools/spmm_microbench.py $wl 2>&1 | grep " L " >> $out
  done
done
