#!/bin/bash
# usage: try_geom.sh <config> "ns:ranges ..."   (0 = library default)
c=$1; shift
for v in "$@"; do
  ns=${v%%:*}; nr=${v##*:}
  echo -n "c$c ns=$ns ranges=$nr: "
  DPGP_PSI2_NS=$ns DPGP_PP_RANGES=$nr python bench.py --config $c --steps 40 --warmup 5 --no-cpu-baseline --no-secondary --no-grad --no-side 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['roofline']['kernel_ms'],4), round(d['roofline']['exp_frac'],3))"
done
