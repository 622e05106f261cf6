#!/bin/bash
# scratch: forward time at a config for n-splits of the Psi1^T y kernel
for ns in 1 2 3 4 6 8 16; do
  export DPGP_PSI1_NS=$ns
  printf "cfg %s ns1=%d " $1 $ns
  timeout -k 10 120 python3 bench.py --config $1 --no-cpu-baseline --no-secondary --no-grad 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f ms' % d['ms_per_step'])"
done
