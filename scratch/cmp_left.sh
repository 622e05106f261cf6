#!/bin/bash
# the left-looking persistent Cholesky against the right-looking one (DPGP_POTRF_LEFT=0): correctness (scratch/persist_check.py shapes) and time at B = 256, M = 512
cd "$(dirname "$0")/.."
for left in 1 0; do
  echo "== DPGP_POTRF_LEFT=$left"
  DPGP_POTRF_LEFT=$left timeout -k 10 120 python scratch/persist_check.py 1
done
