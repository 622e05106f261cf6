#!/bin/bash
# quick kernel-trace stats of the bench for one config: scratch/trace_quick.sh <config> [extra bench flags]
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/trace_quick_$1; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $R
c=$1; shift
rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -- python3 bench.py --config $c --steps 30 --warmup 5 --no-cpu-baseline --no-secondary --no-grad --no-side "$@" > $O/log.txt 2>&1
python3 - <<PY
import csv, glob
for f in glob.glob("$O/t/*/*kernel_stats.csv"):
    for r in list(csv.DictReader(open(f)))[:10]:
        print("  %-60s calls %4s avg %10.1f us  %6s%%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
find $O -name "*kernel_trace.csv" -delete
