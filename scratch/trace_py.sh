#!/bin/bash
# usage: trace_py.sh <tag> <script.py> [args]: wall time of the script, then rocprofv3 kernel-trace stats of it -> gpurun_out/trace_<tag>
tag=$1; shift
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/trace_$tag
rm -rf $O; mkdir -p $O
cd $R
python3 "$@" 2>/dev/null
cd /tmp && export TMPDIR=/tmp
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 "$@" > $O/log.txt 2>&1
find $O -name "*kernel_trace.csv" -delete
f=$(find $O -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('kernel time total %.1f ms over %d kernel names, %d launches' % (tot / 1e6, len(rows), sum(int(r['Calls']) for r in rows)))
for r in rows[:22]:
    print(r['Name'][:64].ljust(64), r['Calls'].rjust(5), '%9.1f us avg %6.2f%%' % (float(r['AverageNs']) / 1e3, 100 * float(r['TotalDurationNs']) / tot))
PY
