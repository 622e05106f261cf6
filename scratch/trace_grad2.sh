#!/bin/bash
# kernel-trace stats of the MIXED gradient path only (no f64-forward run, no over-T side model): scratch/trace_grad2.sh <config>
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/trace_grad2_$1; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $R
cat > $O/run.py <<PY
import sys, numpy as np, torch
sys.path.insert(0, "$R")
from dp_gp_lvm_amd.models.dp_gp_lvm import dp_gp_lvm
from dp_gp_lvm_amd.utils.synthetic import make_problem, CONFIGS
c = int(sys.argv[1]); n, d, m, q = CONFIGS[c]; p = make_problem(c); t = p['phi'].shape[1]
init = dict(x_mean=p['mu'], x_var=p['s'], x_u=p['z'], phi_logits=np.log(p['phi']), gamma_atoms=p['gamma_atoms'], alpha_atoms=p['alpha_atoms'], beta_atoms=p['beta_atoms'], gamma_1=p['g1'], gamma_2=p['g2'], w_1=p['w1'], w_2=p['w2'])
model = dp_gp_lvm(p['y'], num_latent_dims=q, num_inducing_points=m, truncation_level=t, alpha_prior_params=np.array([p['s1'], p['s2']]), device='cuda:0', initial_values=init, precision='mixed')
for _ in range(6): model.gradients()
torch.cuda.synchronize()
PY
rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -- python3 $O/run.py $1 > $O/log.txt 2>&1
python3 - <<PY
import csv, glob
for f in glob.glob("$O/t/*/*kernel_stats.csv"):
    for r in list(csv.DictReader(open(f)))[:22]:
        print("  %-64s calls %4s avg %9.1f us   per iteration %7.3f ms" % (r["Name"][:64], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6/6))
PY
find $O -name "*kernel_trace.csv" -delete
