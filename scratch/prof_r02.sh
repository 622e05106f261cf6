#!/bin/bash
# Round-2 profile set -> gpurun_out/prof_r02 (copy the summaries to profiles/r02): kernel-trace stats of the default bench and
# of config 2, then the PMC passes (FETCH_SIZE and WRITE_SIZE in separate runs, SQ groups in separate runs), as the
# MI355X guide prescribes; nothing else is traced in the PMC runs.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r02
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R
S="--no-cpu-baseline --no-secondary --no-grad --no-side"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_c3 -- python3 bench.py --steps 40 --warmup 5 $S > $O/bench_trace_c3.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_c2 -- python3 bench.py --steps 40 --warmup 5 --config 2 $S > $O/bench_trace_c2.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_c3_f64 -- python3 bench.py --steps 10 --warmup 3 --prec f64 $S > $O/bench_trace_c3_f64.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 bench.py --steps 4 --warmup 1 $S > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 bench.py --steps 4 --warmup 1 $S > $O/pmc_write.log 2>&1
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $O/pmc1 -- python3 bench.py --steps 4 --warmup 1 $S > $O/pmc1.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_SALU SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmc2 -- python3 bench.py --steps 4 --warmup 1 $S > $O/pmc2.log 2>&1
python3 - <<PY
import csv, glob, collections, os, json, subprocess
O="$O"
out=open(os.path.join(O,"summary_rocprofv3.txt"),"w")
def P(*a):
    s=" ".join(str(x) for x in a); print(s); out.write(s+"\n")
for tag, cmd in (("trace_c3","python3 bench.py --steps 40 --warmup 5"),("trace_c2","python3 bench.py --steps 40 --warmup 5 --config 2"),("trace_c3_f64","python3 bench.py --steps 10 --warmup 3 --prec f64")):
    P("==", tag, "rocprofv3 --kernel-trace --stats --", cmd, "$S")
    for f in glob.glob(O+"/"+tag+"/*/*kernel_stats.csv"):
        for r in list(csv.DictReader(open(f)))[:10]:
            P("  %-58s calls %4s avg %10.1f us  %6s%%" % (r["Name"][:58], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
        os.replace(f, os.path.join(O, tag+"_kernel_stats.csv"))
P("== PMC (python3 bench.py --steps 4 --warmup 1 $S; one rocprofv3 run per line group), averages per dispatch")
vals=collections.defaultdict(dict)
for tag in ("pmc_fetch","pmc_write","pmc1","pmc2"):
    for f in glob.glob(O+"/"+tag+"/*/*counter_collection.csv"):
        agg=collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"].split("(")[0]
            if any(x in k for x in ("psi2_pairs","chain_b","psi1T_y","gram_kernel","kl_yy","psi2_pair_scale")):
                agg[k[:44]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k in agg:
            for c,v in agg[k].items():
                P("  %-46s %-28s %.5g" % (k, c, sum(v)/len(v)))
                vals[k][c]=sum(v)/len(v)
# traffic of the dominant kernel: FETCH_SIZE / WRITE_SIZE are KiB per dispatch; on gfx950 FETCH_SIZE reports half of the bytes
# of wide coalesced reads (MI355X_MICROARCH.md, HBM section): doubled
for k,v in vals.items():
    if "psi2_pairs" in k and "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        rec={"config3_mixed":{"kernel":k.strip(),"fetch_size_kib":v["FETCH_SIZE"],"write_size_kib":v["WRITE_SIZE"],
             "bytes_per_launch":2*1024*v["FETCH_SIZE"]+1024*v["WRITE_SIZE"],
             "rule":"2 x FETCH_SIZE + WRITE_SIZE (KiB -> bytes), separate --pmc passes, MI355X_MICROARCH.md HBM section",
             "command":"rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE -- python3 bench.py --steps 4 --warmup 1 $S"}}
        json.dump(rec, open(os.path.join(O,"traffic.json"),"w"), indent=1)
        P("traffic.json:", json.dumps(rec))
PY
find $O -name "*kernel_trace.csv" -delete
find $O -name "*counter_collection.csv" -delete
find $O -name "*agent_info.csv" -delete
find $O -name "*domain_stats.csv" -delete
du -sh $O
