#!/bin/bash
# kernel-trace stats of the gradient path: scratch/trace_grad.sh <config>
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/trace_grad_$1; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -- python3 bench.py --config $1 --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-side > $O/log.txt 2>&1
python3 - <<PY
import csv, glob
for f in glob.glob("$O/t/*/*kernel_stats.csv"):
    for r in list(csv.DictReader(open(f)))[:16]:
        print("  %-70s calls %4s avg %10.1f us  %6s%%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
find $O -name "*kernel_trace.csv" -delete
