#!/bin/bash
# Round-3 profile set -> gpurun_out/prof_r03 (copy the summaries to profiles/r03): kernel-trace stats of the bench at configs
# 3 (default), 2, 4, 5 and 3 in fp64, the bench lines of the same build, the gram / Cholesky workloads (scratch/prof_linalg.py),
# then the PMC passes — FETCH_SIZE and WRITE_SIZE in separate runs, SQ groups in separate runs, nothing else traced in a PMC
# run (MI355X_MICROARCH.md).  usage: bash scratch/prof_r03.sh [part]   part = trace | pmc | all
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r03
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R
part=${1:-all}
S="--no-cpu-baseline --no-secondary --no-grad --no-side"
if [ "$part" = trace ] || [ "$part" = all ]; then
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_c3 -- python3 bench.py --steps 40 --warmup 5 $S > $O/bench_trace_c3.log 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_c2 -- python3 bench.py --steps 40 --warmup 5 --config 2 $S > $O/bench_trace_c2.log 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_c4 -- python3 bench.py --steps 20 --warmup 3 --config 4 $S > $O/bench_trace_c4.log 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_c5 -- python3 bench.py --steps 40 --warmup 5 --config 5 $S > $O/bench_trace_c5.log 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_c3_f64 -- python3 bench.py --steps 10 --warmup 3 --prec f64 $S > $O/bench_trace_c3_f64.log 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_linalg -- python3 scratch/prof_linalg.py 8 > $O/trace_linalg.log 2>&1
  echo "traces done" > $O/progress.txt
  python3 bench.py > $O/bench_default_output.json 2> $O/bench_default.err
  python3 bench.py --config 2 --no-cpu-baseline > $O/bench_config2_output.json 2>> $O/bench_default.err
  python3 bench.py --config 4 --no-cpu-baseline --steps 10 --warmup 2 > $O/bench_config4_output.json 2>> $O/bench_default.err
  python3 bench.py --config 5 --no-cpu-baseline > $O/bench_config5_output.json 2>> $O/bench_default.err
  echo "bench lines done" >> $O/progress.txt
fi
if [ "$part" = pmc ] || [ "$part" = all ]; then
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 bench.py --steps 4 --warmup 1 $S > $O/pmc_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 bench.py --steps 4 --warmup 1 $S > $O/pmc_write.log 2>&1
  rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $O/pmc1 -- python3 bench.py --steps 4 --warmup 1 $S > $O/pmc1.log 2>&1
  echo "bench pmc done" >> $O/progress.txt
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_la_fetch -- python3 scratch/prof_linalg.py 3 > $O/pmc_la_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_la_write -- python3 scratch/prof_linalg.py 3 > $O/pmc_la_write.log 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_la_sq -- python3 scratch/prof_linalg.py 3 > $O/pmc_la_sq.log 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_c3_sq -- python3 bench.py --steps 4 --warmup 1 $S > $O/pmc_c3_sq.log 2>&1
  echo "linalg pmc done" >> $O/progress.txt
fi
python3 scratch/prof_r03_digest.py $O
find $O -name "*kernel_trace.csv" -delete
find $O -name "*counter_collection.csv" -delete
find $O -name "*agent_info.csv" -delete
find $O -name "*domain_stats.csv" -delete
du -sh $O
