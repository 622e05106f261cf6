#!/bin/bash
# psi2 pair-tile kernel at config 4 (64 output dims): eight waves (product) against one wave per SIMD with G = 4 / 6 resident tiles
cd "$(dirname "$0")/.."
echo "product (NW = 8)"; timeout -k 10 120 python scratch/time_psi2_algo.py 4 64 auto 2>/dev/null
for v in w1g4 w1g6; do
  echo "$v, 4 waves, 160 KB"; DPGP_LIBRARY=scratch/libdpgp_hip_$v.so DPGP_PP_NW=4 DPGP_PP_LDS_KB=160 timeout -k 10 120 python scratch/time_psi2_algo.py 4 64 auto 2>/dev/null
done
