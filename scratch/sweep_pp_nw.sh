#!/bin/bash
# psi2 pair-tile kernel: 4 waves x 2 workgroups per CU against 8 waves x 1 (DPGP_PP_NW), operator alone
cd "$(dirname "$0")/.."
for nw in 4 8; do
  echo "DPGP_PP_NW=$nw"
  DPGP_PP_NW=$nw timeout -k 10 120 python scratch/time_psi2_algo.py 4 64 auto 2>/dev/null
  DPGP_PP_NW=$nw timeout -k 10 120 python scratch/time_psi2_algo.py 5 560 auto 2>/dev/null
  DPGP_PP_NW=$nw timeout -k 10 120 python scratch/time_psi2_algo.py 3 512 auto 2>/dev/null
  DPGP_PP_NW=$nw timeout -k 10 120 python scratch/time_psi2_algo.py 2 64 auto 2>/dev/null
done
