#!/bin/bash
# A/B of an env switch on the config-3 step and the other configurations: ab_env_all.sh OUT "ENV_A" "ENV_B"
out=$1; shift
bash tools/scratch/ab3_bench.sh ${out}_c3 2 "$@" > ${out}_c3.txt 2>&1
{
for rep in 1 2; do
for e in "$@"; do
  for cfg in faust_lap arap_lap; do
    env $e python tools/train_bench.py $cfg 40 2>&1 | grep -E "ms/step" | tail -1 | sed "s/^/[$e] /"
  done
done
done
} > ${out}_other.txt 2>&1
