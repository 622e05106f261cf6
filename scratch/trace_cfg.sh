#!/bin/bash
# kernel-trace stats of the bench at one config -> gpurun_out/trace_cfg<c> (quick look; the judged set is scratch/prof_r03.sh)
c=${1:-3}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/trace_cfg$c
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 bench.py --steps 30 --warmup 3 --config $c --no-cpu-baseline --no-secondary --no-grad --no-side > $O/bench.log 2>&1
find $O -name "*kernel_trace.csv" -delete
f=$(find $O -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if int(r['Calls']) >= 20: print(r['Name'][:60].ljust(60), r['Calls'].rjust(4), '%10.1f us' % (float(r['AverageNs']) / 1e3))
PY
