#!/bin/bash
# On the GPU box: the Laplacian microbench (loop + per-launch events) with each library variant, on the SAME box.
cd "$(dirname "$0")/../.."
cp surfacenetworks_amd/libsn_hip.so /tmp/libsn_default.so
for rep in 1 2; do
for v in "$@"; do
  cp tools/scratch/hints/$v.so surfacenetworks_amd/libsn_hip.so
  echo "== $v (rep $rep)"
  python tools/scratch/ring_timing.py 2>&1 | grep "^ring"
done
done
cp /tmp/libsn_default.so surfacenetworks_amd/libsn_hip.so
