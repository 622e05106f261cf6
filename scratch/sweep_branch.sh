#!/bin/bash
# Psi1^T y on a second stream beside the psi2 launch (DPGP_PARALLEL_BRANCH=1) against the serial stream, configs 3 / 5 / 2
cd "$(dirname "$0")/.."
for c in 3 5 2; do
for pb in 0 1; do
  r=$(DPGP_PARALLEL_BRANCH=$pb timeout -k 10 200 python bench.py --config $c --steps 200 --warmup 20 --no-cpu-baseline --no-secondary --no-grad --no-side 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'])")
  echo "config $c parallel_branch=$pb ms=$r"
done
done
