#!/bin/bash
# usage: build_variant.sh <tag> <extra hipcc flags...>  -> scratch/libdpgp_hip_<tag>.so   (load with DPGP_LIBRARY=...)
# only psi2_pairs.hip is rebuilt with the extra flags; the other objects are the product build's
set -e
tag=$1; shift
cd /root/repo/dp_gp_lvm_amd/csrc
mkdir -p /root/repo/scratch/_v_$tag
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast -mllvm -amdgpu-mfma-vgpr-form "$@" -c psi2_pairs.hip -o /root/repo/scratch/_v_$tag/psi2_pairs.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /root/repo/scratch/libdpgp_hip_$tag.so _build/elementwise.o _build/psi2.o _build/linalg.o _build/potrf_big.o _build/gemm.o _build/grad.o _build/elbo.o /root/repo/scratch/_v_$tag/psi2_pairs.o
