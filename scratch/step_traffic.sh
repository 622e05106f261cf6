#!/bin/bash
# HBM traffic of the training step's kernels at one config: FETCH_SIZE and WRITE_SIZE in separate --pmc passes (KiB per dispatch; FETCH_SIZE is
# doubled on gfx950: MI355X_MICROARCH.md), kernel-trace durations from a third run -> gpurun_out/step_traffic_<config>/step_traffic.json
# usage: scratch/step_traffic.sh <config>
R=$GRAFT_REPO_ROOT; C=${1:-3}; O=$R/gpurun_out/step_traffic_$C; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $R
sed -n '/^cat > \$O\/run.py/,/^PY$/p' scratch/grad_pmc.sh | sed '1d;$d' | sed "s#\$R#$R#g" > $O/run.py
rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -- python3 $O/run.py $C 6 > $O/log.txt 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/f -- python3 $O/run.py $C 2 >> $O/log.txt 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/w -- python3 $O/run.py $C 2 >> $O/log.txt 2>&1
python3 - <<PY
import csv, glob, json, collections
dur = {}
for f in glob.glob("$O/t/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        dur[r["Name"].split("(")[0][:48]] = (float(r["AverageNs"]) / 1e3, int(r["Calls"]) / 6.0)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for tag in ("f", "w"):
    for f in glob.glob("$O/%s/*/*counter_collection.csv" % tag):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"].split("(")[0][:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, (us, calls) in sorted(dur.items(), key=lambda kv: -kv[1][0] * kv[1][1]):
    c = acc.get(k)
    if not c or us * calls < 20.0: continue
    fe = sum(c.get("FETCH_SIZE", [0])) / max(len(c.get("FETCH_SIZE", [1])), 1); wr = sum(c.get("WRITE_SIZE", [0])) / max(len(c.get("WRITE_SIZE", [1])), 1)
    byt = (2.0 * fe + wr) * 1024.0
    out[k] = {"avg_us": round(us, 1), "launches_per_step": round(calls, 2), "fetch_kib_x2": round(2 * fe), "write_kib": round(wr),
              "hbm_mb_per_launch": round(byt / 1e6, 1), "gb_per_s": round(byt / (us * 1e-6) / 1e9, 1)}
    print("  %-48s %8.1f us x %4.1f  %9.1f MB  %7.1f GB/s" % (k, us, calls, byt / 1e6, byt / (us * 1e-6) / 1e9))
json.dump({"config": $C, "rule": "(2 x FETCH_SIZE + WRITE_SIZE) KiB per dispatch, averaged over the dispatches of a kernel name; durations from a kernel-trace run of the same script", "kernels": out}, open("$O/step_traffic.json", "w"), indent=1)
PY
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*agent_info.csv" -delete
