#!/bin/bash
# ms per evaluation and psi2 kernel ms at the given configs (default 3 2 5): three runs each
cd "$(dirname "$0")/.."
for c in ${@:-3 2 5}; do
  for i in 1 2 3; do
    st=200; [ $c = 4 ] && st=8
    r=$(timeout -k 10 200 python bench.py --config $c --steps $st --warmup 20 --no-cpu-baseline --no-secondary --no-grad --no-side 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%.4f ms, psi2 %.4f ms, frac %.3f' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac']))")
    echo "config $c: $r"
  done
done
