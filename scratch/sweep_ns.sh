#!/bin/bash
# scratch: forward time at a config for every n-split of the psi2 dispatch
for ns in 1 2 3 4 5 6 7 8; do
  export DPGP_PSI2_NS=$ns
  printf "ns=%d " $ns
  timeout -k 10 120 python3 bench.py --config $1 --no-cpu-baseline --no-secondary --no-grad 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f ms  psi2 %.4f ms' % (d['ms_per_step'], d['roofline']['kernel_ms']))"
done
