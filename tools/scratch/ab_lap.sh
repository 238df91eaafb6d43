#!/bin/bash
# Laplacian configurations with and without the maxima of the fused transposed Laplacian products (SN_LAP_ABSMAX=0: the weight
# gradients behind them fall back to three bf16 pieces)
for rep in 1 2; do
for e in "SN_LAP_ABSMAX=0" "SN_LAP_ABSMAX=1"; do
  for cfg in arap_lap faust_lap mnist_lap; do
    env $e python tools/train_bench.py $cfg 40 2>&1 | grep -E "ms/step" | tail -1 | sed "s/^/[$e] /"
  done
done
done
