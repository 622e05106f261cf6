#!/bin/bash
# diagnostic build of the library with -DDPGP_PROFILE_CHAIN into scratch/libdpgp_hip_dbg.so
set -e
cd /root/repo/dp_gp_lvm_amd/csrc
mkdir -p /root/repo/scratch/_dbg
for f in elementwise psi2 linalg potrf_big grad elbo; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast -mllvm -amdgpu-mfma-vgpr-form -DDPGP_PROFILE_CHAIN -c $f.hip -o /root/repo/scratch/_dbg/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /root/repo/scratch/libdpgp_hip_dbg.so /root/repo/scratch/_dbg/*.o
