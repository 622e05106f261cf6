#!/bin/bash
# over-T model at config 3: gradients() eager / from the HIP graph / one optimise() iteration, then kernel-trace stats of the same -> gpurun_out/trace_tstep
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/trace_tstep; rm -rf $O; mkdir -p $O
cd $R; python3 scratch/time_model_t_step.py 3 2>/dev/null
cd /tmp && export TMPDIR=/tmp; cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -- python3 scratch/time_model_t_step.py 3 > $O/log.txt 2>&1
python3 - <<PY
import csv, glob
for f in glob.glob("$O/t/*/*kernel_stats.csv"):
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows); calls = sum(int(r["Calls"]) for r in rows)
    print("  total kernel time %.1f ms in %d launches" % (tot / 1e6, calls))
    for r in rows[:22]:
        print("  %-60s calls %5s avg %8.1f us  total %8.2f ms" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6))
PY
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
