#!/bin/bash
# config 2 (D = 64) over (n-splits, tile ranges) of the pair-tile kernel: ms per evaluation
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
out=gpurun_out/sweep_c2.txt; : > $out
for c in "0 0" "7 1" "4 2" "3 2" "2 3" "2 4" "1 7" "4 1" "5 1" "6 1" "8 1" "3 3"; do
  set -- $c
  if [ "$1" = 0 ]; then unset DPGP_PSI2_NS DPGP_PP_RANGES; else export DPGP_PSI2_NS=$1 DPGP_PP_RANGES=$2; fi
  r=$(timeout -k 10 120 python bench.py --config 2 --steps 400 --warmup 40 --no-cpu-baseline --no-secondary --no-grad --no-side 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'])")
  echo "ns=$1 nr=$2 ms=$r" | tee -a $out
done
