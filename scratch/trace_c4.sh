#!/bin/bash
# kernel-trace stats of the config-4 bench -> gpurun_out/trace_c4 (quick look; the judged set is scratch/prof_r03.sh)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/trace_c4
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 bench.py --steps 20 --warmup 3 --config 4 --no-cpu-baseline --no-secondary --no-grad --no-side > $O/bench.log 2>&1
find $O -name "*kernel_trace.csv" -delete
find $O -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'cut -d, -f1-5 {} | head -20'
