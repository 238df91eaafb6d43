#!/bin/bash
# Ablation builds of the whole library with wgrad_u_k switches (tools/scratch/hints/<name>.so), for tools/scratch/wgrad_ab.sh
cd "$(dirname "$0")/../.."
mkdir -p tools/scratch/hints
src="surfacenetworks_amd/csrc/sn_kernels.hip surfacenetworks_amd/csrc/sn_dense.hip surfacenetworks_amd/csrc/sn_meshops.hip surfacenetworks_amd/csrc/sn_gemm.hip"
build() { name=$1; shift; /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -I include "$@" $src -o tools/scratch/hints/$name.so & }
build wgu_mfma_only -DSN_X_WGU_NOLOAD=1 -DSN_X_WGU_NOCONV=1
build wgu_nomfma -DSN_X_WGU_NOMFMA=1
build wgu_half -DSN_X_WGU_HALF=1
build wgu_half_noconv -DSN_X_WGU_HALF=1 -DSN_X_WGU_NOCONV=1
build wgu_noconv -DSN_X_WGU_NOCONV=1
build wgu_loads_only -DSN_X_WGU_NOMFMA=1 -DSN_X_WGU_NOCONV=1
wait
ls -la tools/scratch/hints/wgu_*.so
