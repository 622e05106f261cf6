#!/bin/bash
# kernel-trace stats only (configs 3 and 2). usage: prof_trace.sh <tag>
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/trace_$1
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R
for c in 3 2; do
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c$c -- python3 bench.py --steps 40 --warmup 5 --config $c --no-cpu-baseline --no-secondary > $O/bench_c$c.log 2>&1
done
python3 - <<PY
import csv, glob
for c in (3, 2):
    print("== config", c)
    for f in glob.glob("$O/c%d/*/*kernel_stats.csv" % c):
        for r in list(csv.DictReader(open(f)))[:9]:
            print("  %-48s calls %4s avg %9.1f us  %6s%%" % (r["Name"].replace("void ", "")[:48], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
    for l in open("$O/bench_c%d.log" % c):
        if l.startswith("{"): print("  ", l[:150])
PY
