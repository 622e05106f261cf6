#!/bin/bash
# over-T objective at config 3: wall time, then the kernel-trace stats of the same loop -> gpurun_out/trace_t
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/trace_t
mkdir -p $O
cd $R
python3 scratch/time_model_t.py 3 2>/dev/null
DPGP_FUSED_T=0 python3 scratch/time_model_t.py 3 2>/dev/null
cd /tmp && export TMPDIR=/tmp
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 scratch/time_model_t.py 3 > $O/log.txt 2>&1
find $O -name "*kernel_trace.csv" -delete
