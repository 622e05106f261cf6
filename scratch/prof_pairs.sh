#!/bin/bash
# PMC passes on the default bench (psi2 pair-tile kernel): separate runs for the SQ groups, as gpurun requires
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2h
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R
B="python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-secondary --no-grad --no-side"
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $O/pmc1 -- $B > $O/pmc1.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_SALU SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmc2 -- $B > $O/pmc2.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-grad --no-side > $O/trace.log 2>&1
python3 - <<PY
import csv, glob, collections, os
O="$O"
out=open(os.path.join(O,"summary.txt"),"w")
def P(*a):
    s=" ".join(str(x) for x in a); print(s); out.write(s+"\n")
for f in glob.glob(O+"/trace/*/*kernel_stats.csv"):
    for r in list(csv.DictReader(open(f)))[:9]:
        P("  %-58s calls %4s avg %10.1f us  %6s%%" % (r["Name"][:58], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
for tag in ("pmc1","pmc2"):
    for f in glob.glob(O+"/"+tag+"/*/*counter_collection.csv"):
        agg=collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"].split("(")[0]
            if any(x in k for x in ("psi2_pairs","psi2_f16")):
                agg[k[:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k in agg:
            for c,v in agg[k].items():
                P("  %-42s %-28s %.5g" % (k, c, sum(v)/len(v)))
PY
find $O -name "*kernel_trace.csv" -delete
find $O -name "*counter_collection.csv" -size +8M -delete
