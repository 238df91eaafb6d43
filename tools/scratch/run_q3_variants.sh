#!/bin/bash
# On the GPU box: config-5 / config-3 Dirac microbench with each q3 kernel variant (same box).  Restores the default library.
cd "$(dirname "$0")/../.."
cp surfacenetworks_amd/libsn_hip.so /tmp/libsn_default.so
out=${1:-gpurun_out/q3_variants.txt}; shift
: > $out
for v in default "$@"; do
  [ "$v" = default ] && cp /tmp/libsn_default.so surfacenetworks_amd/libsn_hip.so || cp tools/scratch/hints/$v.so surfacenetworks_amd/libsn_hip.so
  for wl in c5 c3; do
    echo "== $v $wl" >> $out
    SN_MB_ONLY=bsr4 python tools/spmm_microbench.py $wl 2>&1 | grep " q3 " >> $out
  done
done
cp /tmp/libsn_default.so surfacenetworks_amd/libsn_hip.so
