#!/bin/bash
# On the GPU box: the weight-gradient probe with each kernel variant (and each library variant of tools/scratch/hints/) on the SAME box.
cd "$(dirname "$0")/../.."
cp surfacenetworks_amd/libsn_hip.so /tmp/libsn_default.so
for rep in 1 2; do
for wv in 2 1; do SN_WGRAD_VARIANT=$wv python tools/scratch/wgrad_probe.py variant$wv; done
for v in "$@"; do
  cp tools/scratch/hints/$v.so surfacenetworks_amd/libsn_hip.so
  python tools/scratch/wgrad_probe.py $v
done
cp /tmp/libsn_default.so surfacenetworks_amd/libsn_hip.so
done
