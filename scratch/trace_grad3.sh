#!/bin/bash
# kernel-trace stats of model.gradients() (mixed; the fused training step by default) at one config: scratch/trace_grad3.sh <config> [tag]
R=$GRAFT_REPO_ROOT; C=${1:-3}; O=$R/gpurun_out/trace_g_$C$2; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $R
sed -n '/^cat > \$O\/run.py/,/^PY$/p' scratch/grad_pmc.sh | sed '1d;$d' | sed "s#\$R#$R#g" > $O/run.py
rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -- python3 $O/run.py $C 6 > $O/log.txt 2>&1
python3 - <<PY
import csv, glob
for f in glob.glob("$O/t/*/*kernel_stats.csv"):
    tot = 0.0
    rows = list(csv.DictReader(open(f)))
    for r in rows: tot += float(r["TotalDurationNs"])
    print("  total kernel time per iteration %.3f ms" % (tot / 1e6 / 6))
    for r in rows[:int("${3:-26}")]:
        print("  %-64s calls %4s avg %9.1f us   per iteration %7.3f ms" % (r["Name"][:64], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6/6))
PY
mkdir -p $O/keep; cp $O/t/*/*kernel_stats.csv $O/keep/ 2>/dev/null; find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
