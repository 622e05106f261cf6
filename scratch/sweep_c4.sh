#!/bin/bash
# config 4 over the n-splits of the pair-tile kernel (eight-wave workgroups): ms per evaluation, psi2 kernel ms
cd "$(dirname "$0")/.."
for ns in 0 1 2 3 4; do
  if [ "$ns" = 0 ]; then unset DPGP_PSI2_NS; else export DPGP_PSI2_NS=$ns; fi
  r=$(timeout -k 10 200 python bench.py --config 4 --steps 8 --warmup 2 --no-cpu-baseline --no-secondary --no-grad --no-side 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['roofline']['kernel_ms'])")
  echo "ns=$ns ms=$r"
done
