#!/bin/bash
# Gradient path (mixed) at one config: kernel-trace stats, then SQ counter passes per kernel (separate --pmc runs, nothing else traced)
# usage: scratch/grad_pmc.sh <config> [tag]   -> gpurun_out/grad_pmc_<config><tag>/{trace_stats.txt,pmc.txt}
R=$GRAFT_REPO_ROOT; C=${1:-3}; O=$R/gpurun_out/grad_pmc_$C$2; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $R
cat > $O/run.py <<PY
import sys, os, numpy as np, torch
sys.path.insert(0, "$R")
from dp_gp_lvm_amd.models.dp_gp_lvm import dp_gp_lvm
from dp_gp_lvm_amd.utils.synthetic import make_problem, CONFIGS
c = int(sys.argv[1]); reps = int(sys.argv[2]); n, d, m, q = CONFIGS[c]; p = make_problem(c); t = p['phi'].shape[1]
init = dict(x_mean=p['mu'], x_var=p['s'], x_u=p['z'], phi_logits=np.log(p['phi']), gamma_atoms=p['gamma_atoms'], alpha_atoms=p['alpha_atoms'], beta_atoms=p['beta_atoms'], gamma_1=p['g1'], gamma_2=p['g2'], w_1=p['w1'], w_2=p['w2'])
model = dp_gp_lvm(p['y'], num_latent_dims=q, num_inducing_points=m, truncation_level=t, alpha_prior_params=np.array([p['s1'], p['s2']]), device='cuda:0', initial_values=init, precision=os.environ.get('GP_PREC', 'mixed'), backward_precision='mixed')
for _ in range(reps): model.gradients()
torch.cuda.synchronize()
PY
rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -- python3 $O/run.py $C 6 > $O/log.txt 2>&1
python3 - > $O/trace_stats.txt <<PY
import csv, glob
for f in glob.glob("$O/t/*/*kernel_stats.csv"):
    for r in list(csv.DictReader(open(f)))[:30]:
        print("  %-64s calls %4s avg %9.1f us   per iteration %7.3f ms" % (r["Name"][:64], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6/6))
PY
cat $O/trace_stats.txt
mkdir -p $O/keep; cp $O/t/*/*kernel_stats.csv $O/keep/ 2>/dev/null
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $O/p1 -- python3 $O/run.py $C 2 >> $O/log.txt 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE --output-format csv -d $O/p2 -- python3 $O/run.py $C 2 >> $O/log.txt 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAIT_INST_LDS SQ_INSTS_LDS --output-format csv -d $O/p3 -- python3 $O/run.py $C 2 >> $O/log.txt 2>&1
python3 - > $O/pmc.txt <<PY
import csv, glob, collections
for pas in ("p1", "p2", "p3"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for f in glob.glob("$O/%s/*/*counter_collection.csv" % pas):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:60]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[(k, r["Counter_Name"])] += 1
    for k in sorted(acc, key=lambda k: -acc[k].get("SQ_WAVE_CYCLES", acc[k].get("GRBM_GUI_ACTIVE", 0)))[:12]:
        print(pas, k)
        for c, v in acc[k].items():
            print("      %-32s %.4g per dispatch (%d dispatches)" % (c, v / cnt[(k, c)], cnt[(k, c)]))
PY
cat $O/pmc.txt
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*agent_info.csv" -delete
