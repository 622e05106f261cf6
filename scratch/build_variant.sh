#!/bin/bash
# usage: build_variant.sh <tag> <extra hipcc flags...>  -> scratch/libdpgp_hip_<tag>.so
set -e
tag=$1; shift
cd /root/repo/dp_gp_lvm_amd/csrc
mkdir -p /root/repo/scratch/_v_$tag
for f in elementwise psi2 linalg potrf_big grad elbo; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast -mllvm -amdgpu-mfma-vgpr-form "$@" -c $f.hip -o /root/repo/scratch/_v_$tag/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /root/repo/scratch/libdpgp_hip_$tag.so /root/repo/scratch/_v_$tag/*.o
