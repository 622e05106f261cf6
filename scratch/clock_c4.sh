#!/bin/bash
# effective shader clock of the psi2 dispatch: GRBM_GUI_ACTIVE / duration   usage: clock_c4.sh <config>
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/clock_$1; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $R
S="--no-cpu-baseline --no-secondary --no-grad --no-side"
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/p -- python3 bench.py --config $1 --steps 3 --warmup 1 $S > $O/log.txt 2>&1
python3 - <<PY
import csv, glob, collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/p/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "psi2_pairs" in r["Kernel_Name"]:
            agg[r["Counter_Name"]]["v"].append(float(r["Counter_Value"]))
            if "Start_Timestamp" in r: agg["_dur"]["v"].append(float(r["End_Timestamp"])-float(r["Start_Timestamp"]))
for k,v in agg.items(): print(k, sum(v["v"])/len(v["v"]))
PY
for f in $O/p/*/*kernel_trace.csv; do python3 - <<PY
import csv
d=[(float(r["End_Timestamp"])-float(r["Start_Timestamp"])) for r in csv.DictReader(open("$f")) if "psi2_pairs" in r["Kernel_Name"]]
print("kernel_trace durations ns:", d)
PY
done
find $O -name "*.csv" -size +1M -delete
