#!/bin/bash
# psi2 pair-tile kernel at Q = 20 against the LDS budget of a workgroup (rows per chunk of the A image): config 4 (64 output dims) and 5
cd "$(dirname "$0")/.."
for kb in 80 40 56 160; do
  echo "DPGP_PP_LDS_KB=$kb"
  DPGP_PP_LDS_KB=$kb timeout -k 10 120 python scratch/time_psi2_algo.py 4 64 auto 2>/dev/null
  DPGP_PP_LDS_KB=$kb timeout -k 10 120 python scratch/time_psi2_algo.py 5 560 auto 2>/dev/null
done
