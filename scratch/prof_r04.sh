#!/bin/bash
# Round-4 profile set -> gpurun_out/prof_r04 (copy the summaries to profiles/r04): kernel-trace stats of the bench at configs
# 3 (default), 2, 4, 5 and 3 in fp64, the bench lines of the same build, the gram / Cholesky workloads (scratch/prof_linalg.py),
# then the PMC passes — FETCH_SIZE and WRITE_SIZE in separate runs, SQ groups in separate runs, nothing else traced in a PMC
# run (MI355X_MICROARCH.md).  usage: bash scratch/prof_r04.sh [part]   part = trace | pmc | all
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r04
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
  # the training step (model.gradients(): dpgp_elbo_step + dpgp_model_backward) at configs 3, 2, 5, 4
  cat > $O/grad_run.py <<PY
import sys, os, numpy as np, torch
sys.path.insert(0, "$R")
from dp_gp_lvm_amd.models.dp_gp_lvm import dp_gp_lvm
from dp_gp_lvm_amd.utils.synthetic import make_problem, CONFIGS
c = int(sys.argv[1]); reps = int(sys.argv[2]); n, d, m, q = CONFIGS[c]; p = make_problem(c); t = p['phi'].shape[1]
init = dict(x_mean=p['mu'], x_var=p['s'], x_u=p['z'], phi_logits=np.log(p['phi']), gamma_atoms=p['gamma_atoms'], alpha_atoms=p['alpha_atoms'], beta_atoms=p['beta_atoms'], gamma_1=p['g1'], gamma_2=p['g2'], w_1=p['w1'], w_2=p['w2'])
model = dp_gp_lvm(p['y'], num_latent_dims=q, num_inducing_points=m, truncation_level=t, alpha_prior_params=np.array([p['s1'], p['s2']]), device='cuda:0', initial_values=init, precision='mixed')
for _ in range(reps): model.gradients()
torch.cuda.synchronize()
PY
  for c in 3 2 5; do rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_grad_c$c -- python3 $O/grad_run.py $c 8 > $O/trace_grad_c$c.log 2>&1; done
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_grad_c4 -- python3 $O/grad_run.py 4 3 > $O/trace_grad_c4.log 2>&1
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
  # the pass kernel of stage B (config 3 training step): the counters the round-3 verdict asked for, two SQ passes + GRBM
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $O/pmc_grad1 -- python3 $O/grad_run.py 3 2 > $O/pmc_grad1.log 2>&1
  rocprofv3 --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_grad2 -- python3 $O/grad_run.py 3 2 > $O/pmc_grad2.log 2>&1
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $O/pmc_grad5_1 -- python3 $O/grad_run.py 5 2 > $O/pmc_grad5_1.log 2>&1
  rocprofv3 --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_grad5_2 -- python3 $O/grad_run.py 5 2 > $O/pmc_grad5_2.log 2>&1
  echo "bench pmc done" >> $O/progress.txt
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_la_fetch -- python3 scratch/prof_linalg.py 3 > $O/pmc_la_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_la_write -- python3 scratch/prof_linalg.py 3 > $O/pmc_la_write.log 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_la_sq -- python3 scratch/prof_linalg.py 3 > $O/pmc_la_sq.log 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_c3_sq -- python3 bench.py --steps 4 --warmup 1 $S > $O/pmc_c3_sq.log 2>&1
  echo "linalg pmc done" >> $O/progress.txt
fi
python3 scratch/prof_r04_digest.py $O
find $O -name "*kernel_trace.csv" -delete
find $O -name "*counter_collection.csv" -delete
find $O -name "*agent_info.csv" -delete
find $O -name "*domain_stats.csv" -delete
du -sh $O
