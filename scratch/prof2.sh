#!/bin/bash
# PMC passes for the psi2 kernel (bench config 3, few steps). usage: prof2.sh <tag>
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$1
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $O/pmc1 -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline > $O/bench_pmc1.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_WAVES SQ_INST_CYCLES_VMEM --output-format csv -d $O/pmc2 -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline > $O/bench_pmc2.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_IFETCH SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT --output-format csv -d $O/pmc3 -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline > $O/bench_pmc3.log 2>&1
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob('$O/pmc*/*/*counter_collection.csv')):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'psi2_f16' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
    for c, v in agg.items():
        print('%-28s %.4g' % (c, sum(v)/len(v)))
PY
