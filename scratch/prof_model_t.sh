#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/model_t
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -- python3 scratch/time_model_t.py $1 > $O/log.txt 2>&1
grep cfg $O/log.txt
python3 - <<PY
import csv, glob
for f in glob.glob("$O/t/*/*kernel_stats.csv"):
    rows=list(csv.DictReader(open(f)))
    tot=sum(float(r["TotalDurationNs"]) for r in rows)
    print("total kernel time per evaluation %.1f us over %d kernels" % (tot/23/1e3, sum(int(r["Calls"]) for r in rows)/23))
    for r in rows[:14]:
        print("  %-70s calls %4s avg %9.1f us" % (r["Name"].replace("void ", "")[:70], r["Calls"], float(r["AverageNs"])/1e3))
PY
