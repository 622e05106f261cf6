#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/grad_$1
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -- python3 scratch/time_grad.py $2 > $O/log.txt 2>&1
grep cfg $O/log.txt
python3 - <<PY
import csv, glob
for f in glob.glob("$O/t/*/*kernel_stats.csv"):
    for r in list(csv.DictReader(open(f)))[:9]:
        print("  %-56s calls %4s avg %9.1f us total %9.1f us" % (r["Name"].replace("void ", "")[:56], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e3))
PY
