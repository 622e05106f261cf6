#!/bin/bash
# psi2 operator alone at config 5 (M = 64, Q = 20: 65 pair tiles) over n-splits and waves per workgroup
cd "$(dirname "$0")/.."
for nw in 4 8; do for ns in 8 4 2 1; do
  echo -n "NW=$nw ns=$ns: "; DPGP_PP_NW=$nw DPGP_PSI2_NS=$ns timeout -k 10 120 python scratch/time_psi2_algo.py 5 560 auto 2>/dev/null | cut -d: -f2
done; done
