#!/bin/bash
# On the GPU box: bench each variant on the SAME box (box-to-box variance is +-3 %).  Restores the default library.
cd "$(dirname "$0")/../.."
cp surfacenetworks_amd/libsn_hip.so /tmp/libsn_default.so
for rep in 1 2; do
for v in "$@"; do
  cp tools/scratch/hints/$v.so surfacenetworks_amd/libsn_hip.so
  python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('%-16s %8.1f meshes/s  %7.3f ms/step  spmm frac %.3f' % ('$v', d['value'], d['ms_per_step'], d['roofline']['frac']))"
done
done
cp /tmp/libsn_default.so surfacenetworks_amd/libsn_hip.so
