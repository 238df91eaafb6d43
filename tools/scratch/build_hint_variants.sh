#!/bin/bash
# Build libsn_hip variants with different cache-hint policies into tools/scratch/hints/<name>.so (experiment only).
# usage: build_hint_variants.sh name1:"-DFLAG=1 -DFLAG2=1" name2:...
set -e
cd "$(dirname "$0")/../.."
SRC="surfacenetworks_amd/csrc/sn_kernels.hip surfacenetworks_amd/csrc/sn_dense.hip surfacenetworks_amd/csrc/sn_meshops.hip surfacenetworks_amd/csrc/sn_gemm.hip"
for spec in "$@"; do
  name="${spec%%:*}"; flags="${spec#*:}"; [ "$flags" = "$spec" ] && flags=""
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -I include $flags $SRC -o tools/scratch/hints/$name.so &
done
wait
ls tools/scratch/hints/
