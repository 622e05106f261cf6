#!/bin/bash
# usage: try_variants.sh "<cfgs>" tag:ldskb ...
cfgs=$1; shift
for v in "$@"; do
  tag=${v%%:*}; kb=${v##*:}
  for c in $cfgs; do
    steps=30; [ $c = 4 ] && steps=6
    if [ $tag = base ]; then lib=; else lib=/root/repo/scratch/libdpgp_hip_$tag.so; fi
    echo -n "$tag kb=$kb c$c: "
    DPGP_LIBRARY=$lib DPGP_PP_LDS_KB=$kb python bench.py --config $c --steps $steps --warmup 3 --no-cpu-baseline --no-secondary --no-grad --no-side 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],2), round(d['roofline']['kernel_ms'],4), round(d['roofline']['exp_frac'],3), d.get('precision_check'))"
  done
done
