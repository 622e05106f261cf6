#!/bin/bash
# Round-1 profile set for profiles/r01: kernel-trace stats of the default bench + PMC passes (separate runs, as required)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_final
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_c3 -- python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-secondary --no-grad > $O/bench_trace_c3.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_c2 -- python3 bench.py --steps 40 --warmup 5 --config 2 --no-cpu-baseline --no-secondary --no-grad > $O/bench_trace_c2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-secondary --no-grad > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-secondary --no-grad > $O/pmc_write.log 2>&1
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $O/pmc1 -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-secondary --no-grad > $O/pmc1.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_SALU SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmc2 -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-secondary --no-grad > $O/pmc2.log 2>&1
python3 - <<PY
import csv, glob, collections, os
O="$O"
out=open(os.path.join(O,"summary.txt"),"w")
def P(*a):
    s=" ".join(str(x) for x in a); print(s); out.write(s+"\n")
for tag in ("trace_c3","trace_c2"):
    P("==", tag, "rocprofv3 --kernel-trace --stats (python3 bench.py --steps 40 --warmup 5%s --no-cpu-baseline)" % (" --config 2" if tag.endswith("c2") else ""))
    for f in glob.glob(O+"/"+tag+"/*/*kernel_stats.csv"):
        for r in list(csv.DictReader(open(f)))[:10]:
            P("  %-58s calls %4s avg %10.1f us  %6s%%" % (r["Name"][:58], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
P("== PMC (python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline), averages per dispatch")
for tag in ("pmc_fetch","pmc_write","pmc1","pmc2"):
    for f in glob.glob(O+"/"+tag+"/*/*counter_collection.csv"):
        agg=collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"].split("(")[0]
            if any(x in k for x in ("psi2_f16","chain_b","chain_k","psi1T_y","gram_kernel")):
                agg[k[:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k in agg:
            for c,v in agg[k].items():
                P("  %-42s %-28s %.5g" % (k, c, sum(v)/len(v)))
PY
find $O -name "*kernel_trace.csv" -delete
find $O -name "*counter_collection.csv" -size +8M -delete
du -sh $O
