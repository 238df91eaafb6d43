#!/bin/bash
out=$1; mkdir -p $out
for il in 0 1; do
  SN_GEMM_W8=0 SN_GEMM_INTERLEAVE=$il python tools/scratch/fwd_w8_probe.py "w8=0 interleave=$il"
  SN_GEMM_W8=2 SN_GEMM_INTERLEAVE=$il python tools/scratch/fwd_w8_probe.py "w8=2 interleave=$il"
  SN_WGRAD_INTERLEAVE=$il python tools/scratch/wgrad_h_probe.py "wgrad interleave=$il"
done > $out/probe.txt 2>&1
