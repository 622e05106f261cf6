#!/bin/bash
# forward evaluation at one config / precision: kernel-trace stats and SQ counters of the top kernels
# usage: scratch/pmc_fwd.sh <config> <prec> [tag]
R=$GRAFT_REPO_ROOT; C=${1:-3}; P=${2:-f64}; O=$R/gpurun_out/pmc_fwd_${C}_$P$3; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $R
S="--no-cpu-baseline --no-secondary --no-grad --no-side --config $C --prec $P"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -- python3 bench.py --steps 10 --warmup 2 $S > $O/log.txt 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $O/p1 -- python3 bench.py --steps 3 --warmup 1 $S >> $O/log.txt 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE --output-format csv -d $O/p2 -- python3 bench.py --steps 3 --warmup 1 $S >> $O/log.txt 2>&1
python3 - <<PY
import csv, glob, collections
for f in glob.glob("$O/t/*/*kernel_stats.csv"):
    for r in list(csv.DictReader(open(f)))[:8]:
        print("  %-64s calls %4s avg %9.1f us" % (r["Name"][:64], r["Calls"], float(r["AverageNs"])/1e3))
for pas in ("p1", "p2"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for f in glob.glob("$O/%s/*/*counter_collection.csv" % pas):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:60]; acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
    for k in sorted(acc, key=lambda k: -acc[k].get("SQ_WAVE_CYCLES", acc[k].get("GRBM_GUI_ACTIVE", 0)))[:3]:
        print(pas, k)
        for c, v in acc[k].items(): print("      %-32s %.4g per dispatch (%d)" % (c, v / cnt[(k, c)], cnt[(k, c)]))
PY
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*agent_info.csv" -delete
