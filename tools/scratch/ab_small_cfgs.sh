#!/bin/bash
# A/B of an env switch on the small-batch configurations (replayed steps) and the config-3 step: ab_small_cfgs.sh OUT "ENV_A" "ENV_B"
out=$1; shift
{
for rep in 1 2; do
for e in "$@"; do
  for cfg in mnist_dir faust_lap mnist_lap; do
    env $e python tools/train_bench.py $cfg 40 2>&1 | grep -E "replay|ms" | tail -1 | sed "s/^/[$e] $cfg: /"
  done
done
done
} > $out 2>&1
