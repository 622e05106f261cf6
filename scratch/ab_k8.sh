#!/bin/bash
# A/B: KS = 8 pass kernel with G = 2 resident column tiles per wave and one register set for the exponent operand (scratch/libdpgp_hip_k8g2s.so)
# vs the default build; then the Q > 15 parity tests on the variant
for c in 5; do
  for v in "" k8g2s; do
    for rep in 1 2; do
      echo -n "variant '$v': "
      if [ -z "$v" ]; then timeout -k 10 300 python scratch/time_grad.py $c 2>&1 | grep "gradients"
      else DPGP_LIBRARY=scratch/libdpgp_hip_$v.so timeout -k 10 300 python scratch/time_grad.py $c 2>&1 | grep "gradients"; fi
    done
  done
done
DPGP_LIBRARY=scratch/libdpgp_hip_k8g2s.so timeout -k 10 600 python -m pytest tests/test_gpu_grad.py -q -x -m gpu -k "matrix_pipe or training_step or fast_stage or fused_step" 2>&1 | tail -3
