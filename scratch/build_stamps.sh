#!/bin/bash
# diagnostic build with per-phase clock stamps in chain_b_kernel (-DDPGP_PROFILE_CHAIN) -> scratch/libdpgp_hip_stamps.so
# (load with DPGP_LIBRARY=scratch/libdpgp_hip_stamps.so; read with scratch/chain_stamps.py).  Only linalg.hip is rebuilt.
set -e
cd /root/repo/dp_gp_lvm_amd/csrc
mkdir -p /root/repo/scratch/_v_stamps
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast -mllvm -amdgpu-mfma-vgpr-form -DDPGP_PROFILE_CHAIN -c linalg.hip -o /root/repo/scratch/_v_stamps/linalg.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /root/repo/scratch/libdpgp_hip_stamps.so _build/elementwise.o _build/psi2.o _build/psi2_pairs.o _build/psi2_pairs_grad.o /root/repo/scratch/_v_stamps/linalg.o _build/potrf_big.o _build/potrf_persist.o _build/chain_big.o _build/gemm.o _build/grad.o _build/elbo.o
