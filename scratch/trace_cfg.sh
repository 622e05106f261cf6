#!/bin/bash
# usage: trace_cfg.sh <cfg> <tag>
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_$2
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$2 -- python3 bench.py --steps 30 --warmup 3 --config $1 --no-cpu-baseline > gpurun_out/prof_$2.log 2>&1
python3 - <<PY
import csv,glob
for f in glob.glob("gpurun_out/prof_$2/*/*kernel_stats.csv"):
    for r in list(csv.DictReader(open(f)))[:11]:
        print("%-46s calls %4s avg %9.1f us  %5s%%" % (r["Name"][:46], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
rows=list(csv.DictReader(open(glob.glob("gpurun_out/prof_$2/*/*kernel_trace.csv")[0])))
rows=[r for r in rows if "Fill" not in r["Kernel_Name"]]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
idx=[i for i,r in enumerate(rows) if "model_prepare" in r["Kernel_Name"]]
a=idx[20]; b=idx[21]
t0=int(rows[a]["Start_Timestamp"])
for r in rows[a:b]:
    print("%8.1f %8.1f  %s" % ((int(r["Start_Timestamp"])-t0)/1e3, (int(r["End_Timestamp"])-t0)/1e3, r["Kernel_Name"][:40]))
PY
