#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_c2
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -- python3 bench.py --steps 40 --warmup 5 --config 2 --no-cpu-baseline --no-secondary --no-grad > $O/log.txt 2>&1
python3 - <<PY
import csv, glob
for f in glob.glob("$O/t/*/*kernel_stats.csv"):
    for r in list(csv.DictReader(open(f)))[:9]:
        print("  %-60s calls %4s avg %9.1f us" % (r["Name"].replace("void ", "")[:60], r["Calls"], float(r["AverageNs"])/1e3))
PY
