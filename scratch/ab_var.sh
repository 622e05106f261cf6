#!/bin/bash
# A/B of a variant library against the default build: scratch/ab_var.sh <variant tag> <config> [<config> ...]
v=$1; shift
for c in "$@"; do
  for rep in 1 2; do
    echo -n "default:  "; timeout -k 10 400 python scratch/time_grad.py $c 2>&1 | grep "gradients"
    echo -n "$v: "; DPGP_LIBRARY=scratch/libdpgp_hip_$v.so timeout -k 10 400 python scratch/time_grad.py $c 2>&1 | grep "gradients"
  done
done
