#!/bin/bash
# usage: build_variant.sh <tag> <file.hip> <extra flags> -> scratch/libdpgp_hip_<tag>.so with that one file rebuilt with the flags
# (the other objects come from dp_gp_lvm_amd/csrc/_build: run make first).  Load with DPGP_LIBRARY=scratch/libdpgp_hip_<tag>.so
set -e
tag=$1; file=$2; shift; shift
cd /root/repo/dp_gp_lvm_amd/csrc
mkdir -p /root/repo/scratch/_v_$tag
base=${file%.hip}
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast -mllvm -amdgpu-mfma-vgpr-form "$@" -c $file -o /root/repo/scratch/_v_$tag/$base.o
objs=""
for f in elementwise psi2 psi2_pairs psi2_pairs_grad linalg potrf_big potrf_persist chain_big gemm grad chain_grad_big elbo; do
  if [ "$f" = "$base" ]; then objs="$objs /root/repo/scratch/_v_$tag/$base.o"; else objs="$objs _build/$f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /root/repo/scratch/libdpgp_hip_$tag.so $objs
