#!/bin/bash
# left-looking Cholesky with odd workgroups started late (PL_DEPHASE_SLEEPS x s_sleep 127): B = 256, M = 512
# variants: for n in 2 4 8; do scratch/build_variant.sh dp$n potrf_persist.hip -DPL_DEPHASE_SLEEPS=$n; done
cd "$(dirname "$0")/.."
echo -n "product: "; timeout -k 10 120 python scratch/persist_check.py 1 2>/dev/null | grep "B 256"
for n in 2 4 8; do echo -n "sleeps $n: "; DPGP_LIBRARY=scratch/libdpgp_hip_dp$n.so timeout -k 10 120 python scratch/persist_check.py 1 2>/dev/null | grep "B 256"; done
