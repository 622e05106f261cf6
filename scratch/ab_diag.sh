#!/bin/bash
# timing-only diagnostic builds of the pass kernel (wrong results): no ring fills / no chunk barrier / no LDS operand reads
c=${1:-5}
for v in "" dNOFILL dNOBAR dNOLDS; do
  for rep in 1 2; do
    echo -n "variant '$v': "
    if [ -z "$v" ]; then timeout -k 10 300 python scratch/time_grad.py $c 2>&1 | grep "gradients"
    else DPGP_LIBRARY=scratch/libdpgp_hip_$v.so timeout -k 10 300 python scratch/time_grad.py $c 2>&1 | grep "gradients"; fi
  done
done
