#!/bin/bash
# kernel-trace stats for config 4 and 5. usage: prof_trace4.sh <tag>
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/trace4_$1
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c4 -- python3 bench.py --steps 3 --warmup 1 --config 4 --no-cpu-baseline --no-secondary > $O/bench_c4.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c5 -- python3 bench.py --steps 20 --warmup 2 --config 5 --no-cpu-baseline --no-secondary > $O/bench_c5.log 2>&1
python3 - <<PY
import csv, glob
for c in (4, 5):
    print("== config", c)
    for f in glob.glob("$O/c%d/*/*kernel_stats.csv" % c):
        for r in list(csv.DictReader(open(f)))[:8]:
            print("  %-48s calls %4s avg %9.1f us  %6s%%" % (r["Name"].replace("void ", "")[:48], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
    for l in open("$O/bench_c%d.log" % c):
        if l.startswith("{"): print("  ", l[:150])
PY
