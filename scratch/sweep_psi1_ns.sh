#!/bin/bash
# config 5 / 4 / 3 over the n-splits of the Psi1^T y kernel: ms per evaluation
cd "$(dirname "$0")/.."
for c in 5 3 4; do
for ns in 0 1 2 4 8; do
  if [ "$ns" = 0 ]; then unset DPGP_PSI1_NS; else export DPGP_PSI1_NS=$ns; fi
  st=100; [ $c = 4 ] && st=8
  r=$(timeout -k 10 200 python bench.py --config $c --steps $st --warmup 5 --no-cpu-baseline --no-secondary --no-grad --no-side 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'])")
  echo "config $c psi1 ns=$ns ms=$r"
done
done
