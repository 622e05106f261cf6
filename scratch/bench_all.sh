#!/bin/bash
# scratch: headline metric for the other BASELINE configs (no CPU baseline, no side measurements)
for c in 2 5 4; do
  timeout -k 10 400 python3 bench.py --config $c --no-cpu-baseline --no-secondary --no-grad 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['metric'], '%.1f evals/s  %.4f ms' % (d['value'], d['ms_per_step']))"
done
