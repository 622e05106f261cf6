#!/bin/bash
# A/B: ring fills of the pass kernel through registers (default build) vs LDS-DMA (scratch/libdpgp_hip_dma.so: -DPG_FILL_DMA)
for c in 3 5 2; do
  for rep in 1 2; do
    echo -n "staged: "; timeout -k 10 300 python scratch/time_grad.py $c 2>&1 | grep "gradients"
    echo -n "dma:    "; DPGP_LIBRARY=scratch/libdpgp_hip_dma.so timeout -k 10 300 python scratch/time_grad.py $c 2>&1 | grep "gradients"
  done
done
