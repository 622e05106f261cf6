#!/bin/bash
# front-kernel time by role: variants that skip roles (fr6: only KL/y'y/consts, fr5: only K_uu tiles, fr3: only the scale table)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp; cd $R
for tag in base fr6 fr5 fr3; do
  O=$R/gpurun_out/trace_front_$tag; rm -rf $O; mkdir -p $O
  if [ $tag = base ]; then export DPGP_LIBRARY=; else export DPGP_LIBRARY=$R/scratch/libdpgp_hip_$tag.so; fi
  DPGP_BENCH_NOCHECK=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -- python3 bench.py --config ${1:-3} --steps 20 --warmup 3 --no-cpu-baseline --no-secondary --no-grad --no-side > $O/log.txt 2>&1
  python3 - <<PY
import csv, glob
for f in glob.glob("$O/t/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "elbo_front" in r["Name"]: print("$tag", "elbo_front avg %.1f us" % (float(r["AverageNs"])/1e3))
PY
  find $O -name "*kernel_trace.csv" -delete
done
