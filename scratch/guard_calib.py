"""Calibration run for the conditioning guard: Adam in precision='f64' (streaming backward stage on the
matrix pipe) on a BASELINE config; every `every` iterations the SAME variables are evaluated by a mixed-precision model
and the two objectives, the guard values and the flags are printed.
    python scratch/guard_calib.py [config=2] [iterations=300] [every=10]"""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, '.')
from dp_gp_lvm_amd.models.dp_gp_lvm import dp_gp_lvm
from dp_gp_lvm_amd.utils.synthetic import make_problem

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 300
every = int(sys.argv[3]) if len(sys.argv) > 3 else 10
p = make_problem(cfg)
iv = dict(x_mean=p['mu'], x_var=p['s'], x_u=p['z'], phi_logits=np.log(p['phi']), gamma_atoms=p['gamma_atoms'],
          alpha_atoms=p['alpha_atoms'], beta_atoms=p['beta_atoms'], gamma_1=p['g1'], gamma_2=p['g2'], w_1=p['w1'], w_2=p['w2'])
kw = dict(num_latent_dims=p['mu'].shape[1], num_inducing_points=p['z'].shape[0], truncation_level=p['phi'].shape[1],
          alpha_prior_params=np.array([p['s1'], p['s2']]), device='cuda:0', initial_values=iv)
m64 = dp_gp_lvm(p['y'], precision='f64', backward_precision='mixed', **kw)
mmx = dp_gp_lvm(p['y'], precision='mixed', **kw)
for k, v in mmx.raw.items():
    v.data = m64.raw[k].data                     # same storage
n = p['y'].shape[0]
rows = []


def probe(it):
    if it % every:
        return
    o64 = float(m64.objective)
    g64 = m64.conditioning_guard.clone()
    omx = float(mmx.objective)
    t64, i64 = m64.per_dimension_terms
    tmx, imx = mmx.per_dimension_terms
    dterm = (tmx.sum(1) - t64.sum(1)).abs()
    gmx = mmx.conditioning_guard
    gam = torch.nn.functional.softplus(m64.raw['gamma_atoms'])
    row = dict(it=it, obj64=o64, objmx=omx, rel=abs(omx - o64) / abs(o64), max_dterm=float(dterm.nan_to_num(1e300).max()),
               guard_max=float(gmx.max()), guard_med=float(gmx.median()), guard64_max=float(g64.max()),
               ratio_max=float((dterm / gmx).nan_to_num(0).max()), flagged=int((imx != 0).sum()), info64=int((i64 != 0).sum()),
               gamma_min=float(gam.min()), gamma_max=float(gam.max()))
    ga, gb = m64.gradients(), mmx.gradients()
    for k in ('x_u', 'x_mean', 'gamma_atoms'):
        row['g_' + k] = float(ga[k].abs().max())
        row['dg_' + k] = float((ga[k] - gb[k]).abs().max() / ga[k].abs().max())
    rows.append(row)
    print(json.dumps(row), flush=True)


t0 = time.time()
probe(0)
try:
    m64.optimise(iters, learning_rate=0.01, callback=lambda it: probe(it + 1))
except FloatingPointError as e:
    print('f64 run stopped:', e)
torch.cuda.synchronize()
print('done in %.1f s; N = %d, DPGP_GUARD_REL * N = %.3g' % (time.time() - t0, n, 2e-3 * n))
