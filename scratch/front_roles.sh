#!/bin/bash
# the front launch by role (diagnostic builds FRONT_DIAG_SKIP: wrong results): 3 = scale table only, 5 = K_uu tiles only, 6 = KL / y'y / constants only
cd "$(dirname "$0")/.."
for k in 3 5 6; do
  echo "== FRONT_DIAG_SKIP=$k"
  DPGP_LIBRARY=scratch/libdpgp_hip_fd$k.so GRAFT_REPO_ROOT=$PWD scratch/trace_cfg.sh ${1:-3} 2>/dev/null | grep elbo_front
done
