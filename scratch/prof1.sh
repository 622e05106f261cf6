#!/bin/bash
# kernel-trace stats + two PMC passes of the default bench (config 3)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r1
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_trace.log 2>&1
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $O/pmc1 -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_pmc1.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_WAVES --output-format csv -d $O/pmc2 -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_pmc2.log 2>&1
find $O -name "*.csv" | head -20
