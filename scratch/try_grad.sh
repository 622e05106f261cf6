#!/bin/bash
# usage: try_grad.sh <config> [lib tag ...]   ("base" = product library)
c=$1; shift
for tag in "$@"; do
  if [ $tag = base ]; then lib=; else lib=/root/repo/scratch/libdpgp_hip_$tag.so; fi
  echo -n "$tag c$c: "
  DPGP_LIBRARY=$lib python bench.py --config $c --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-side 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1])['objective_and_gradients']; print({k: round(v,3) for k,v in d.items() if k.endswith('_ms') or k=='ms'})"
done
