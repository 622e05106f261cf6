#!/bin/bash
# profiles/r01/grad_kernel_stats.txt: kernel stats of objective + gradients at configs 3, 2, 5 and of the over-T objective
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/grad_all
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R
for c in 3 2 5; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/c$c -- python3 scratch/time_grad.py $c > $O/log_c$c.txt 2>&1
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/t3 -- python3 scratch/time_model_t.py 3 > $O/log_t3.txt 2>&1
python3 - <<PY
import csv, glob
out=open("$O/summary.txt","w")
def P(s):
    print(s); out.write(s+"\n")
for tag, cmd in (("c3","scratch/time_grad.py 3"),("c2","scratch/time_grad.py 2"),("c5","scratch/time_grad.py 5"),("t3","scratch/time_model_t.py 3")):
    P("== rocprofv3 --kernel-trace --stats -- python3 %s" % cmd)
    for l in open("$O/log_%s.txt" % tag):
        if l.startswith("cfg"): P(l.rstrip())
    for f in glob.glob("$O/"+tag+"/*/*kernel_stats.csv"):
        for r in list(csv.DictReader(open(f)))[:12]:
            P("  %-60s calls %4s avg %9.1f us total %9.1f us" % (r["Name"].replace("void ", "")[:60], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e3))
PY
