#!/bin/bash
# pass kernel of stage B against the row tiles per chunk of its LDS ring (DPGP_PG_NTB: barriers per row tile): objective + gradients, ms
# usage: scratch/sweep_ntb.sh <config> <ntb> [<ntb> ...]
c=$1; shift
for n in "$@"; do echo -n "cfg $c NTb $n: "; DPGP_PG_NTB=$n timeout -k 10 300 python scratch/time_grad.py $c 2>&1 | grep "gradients"; done
