#!/bin/bash
# usage: build_persist_variant.sh <tag> <extra flags> -> scratch/libdpgp_hip_<tag>.so with potrf_persist.hip rebuilt with the flags
set -e
tag=$1; shift
cd /root/repo/dp_gp_lvm_amd/csrc
mkdir -p /root/repo/scratch/_v_$tag
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast -mllvm -amdgpu-mfma-vgpr-form "$@" -c potrf_persist.hip -o /root/repo/scratch/_v_$tag/potrf_persist.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /root/repo/scratch/libdpgp_hip_$tag.so _build/elementwise.o _build/psi2.o _build/psi2_pairs.o _build/psi2_pairs_grad.o _build/linalg.o _build/potrf_big.o /root/repo/scratch/_v_$tag/potrf_persist.o _build/chain_big.o _build/gemm.o _build/grad.o _build/elbo.o
