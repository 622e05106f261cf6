#!/bin/bash
# scratch: objective + gradients time at config 3 with diagnostic builds of the library (wrong results, timing only)
for v in "" nofin nomfma2 noboth; do
  if [ -z "$v" ]; then unset DPGP_LIBRARY; else export DPGP_LIBRARY=$GRAFT_REPO_ROOT/scratch/libdpgp_hip_$v.so; fi
  echo "== variant '$v'"
  timeout -k 10 200 python3 scratch/time_grad.py 3 2>&1 | grep "cfg 3"
done
