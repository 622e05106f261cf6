"""
TEST INFRASTRUCTURE, CONTAINER-ONLY: a fixture at an ILL-CONDITIONED operating point.
Run as ``python oracle/gen_golden_illcond.py`` in the build container (needs /root/reference; never runs on the GPU box).

Adam on the reference's objective drives the ARD weights down (long length scales), the inducing points become redundant
and K_uu approaches singularity (its small eigenvalues approach the 1e-8 jitter).  That is where an fp32 Psi2 stops being a
substitute for the reference's fp64 arithmetic.  The parity tests at the synthetic STARTING points never
see that regime, so this script
  1. runs the reference's own unmodified ``dp_gp_lvm(...)`` constructor (src/models/dp_gp_lvm.py:21-154) under the PyTorch
     stand-in to get its initial point on a small problem (the steering of oracle/gen_golden_grad.py);
  2. trains from there with Adam (lr 0.05) on the oracle restatement's objective (oracle/dpgp_oracle_torch.py, fp64 CPU
     autograd; pinned to the reference by gen_golden_grad.py) until the conditioning-guard bound of the fp32 path (include/dpgp.h) is 10 x its threshold
     (the reference trains the same way: tf.train.AdamOptimizer(...).minimize(objective),
     test/synthetic_data_hard_test.py:143-155);
  3. hands the trained variable VALUES back to the reference's constructor and lets the reference's own graph give the
     objective and ``tf.gradients`` there; checks the NumPy stand-in (same objective) and the oracle restatement
     (objective 1e-9, gradients 1e-6 of the largest entry; much further into the singular regime, cond(K_uu) 3e8, the
     two fp64 formulations themselves differ by 5e-3 in d/dx_u — there is no reference value to pin there);
  4. writes tests/golden/illcond_ref_*.npz: y, the raw variable values, the reference's objective and gradients, the
     condition numbers and the guard bounds (data only).
"""

import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, REPO)

import numpy as np                                                           # noqa: E402
import torch                                                                 # noqa: E402

from oracle import gen_golden_grad as gg                                     # noqa: E402
from oracle import dpgp_oracle_torch as ot                                   # noqa: E402

NAMES = gg.NAMES
CASES = {   # name: ((N, D, M, Q, T, mask_size, seed), Adam lr, max iterations, target = guard bound / (DPGP_GUARD_REL N))
    'illcond_ref_300_8_64_3_T4': ((300, 8, 64, 3, 4, 1, 31), 0.05, 500, 10.0),
}
GUARD_REL = 2.0e-3          # DPGP_GUARD_REL of include/dpgp.h


def kuu_condition(vals, y):
    """Condition numbers of K_uu + jitter I per output dim, the mixed ARD weights, and the conditioning-guard bound of
    include/dpgp.h (DPGP_INFO_ILL_CONDITIONED) restated from the oracle's pieces:
        2^-23 beta |K^-1|_F |Psi2|_F (1 + beta^2/2 v^T B^-1 v),   B = K_uu + beta Psi2,  v = Psi1^T y."""
    raw = {k: torch.tensor(v) for k, v in zip(NAMES, vals)}
    phi = torch.softmax(raw['dp_logits'], dim=-1)
    sp = torch.nn.functional.softplus
    gamma, alpha, beta = phi @ sp(raw['gamma_atoms_raw']), (phi @ sp(raw['alpha_atoms_raw']))[:, 0], \
        (phi @ sp(raw['beta_atoms_raw']))[:, 0]
    k_uu, p2, v = ot.psi_pieces(torch.as_tensor(y), raw['x_u'], raw['x_mean'], sp(raw['x_var_raw']), gamma, alpha)
    ev = torch.linalg.eigvalsh(k_uu)
    kinv = torch.linalg.inv(k_uu)
    l_b = torch.linalg.cholesky(k_uu + beta[:, None, None] * p2)
    c = torch.linalg.solve_triangular(l_b, v[:, :, None], upper=False)[:, :, 0]
    guard = 2.0 ** -23 * beta * torch.linalg.matrix_norm(kinv) * torch.linalg.matrix_norm(p2) * \
        (1.0 + 0.5 * beta * beta * torch.sum(c * c, dim=-1))
    return (ev[:, -1] / ev[:, 0]).numpy(), gamma.numpy(), guard.numpy()


def main():
    for name, (case, lr, max_it, target) in CASES.items():
        tf, model, variables, y = gg.build_reference_model('standin_torch', case)
        vals = [v.detach().numpy().copy() for v in variables]
        obj0 = float(model.objective)
        # ---- Adam on the restatement's objective (CPU autograd) ----
        params = [torch.tensor(v, dtype=torch.float64, requires_grad=True) for v in vals]
        opt = torch.optim.Adam(params, lr=lr)
        yt = torch.as_tensor(y, dtype=torch.float64)
        it = 0
        for it in range(max_it):
            opt.zero_grad()
            obj, _ = ot.objective(yt, dict(zip(NAMES, params)), s_1=1.0, s_2=1.0, mask_size=case[5])
            obj.backward()
            opt.step()
            if it % 10 == 9:
                cond, gamma, guard = kuu_condition([p.detach().numpy() for p in params], y)
                print('  it %3d objective %.6f  cond(K_uu) max %.3g  gamma %.3g..%.3g  guard / (GUARD_REL N) %.3g' %
                      (it + 1, float(obj), cond.max(), gamma.min(), gamma.max(), guard.max() / (GUARD_REL * y.shape[0])))
                if guard.max() > target * GUARD_REL * y.shape[0]:
                    break
        vals = [p.detach().numpy().copy() for p in params]
        cond, gamma, guard = kuu_condition(vals, y)
        assert guard.max() > target * GUARD_REL * y.shape[0], 'training did not reach the ill-conditioned regime'
        # ---- the reference's own graph at the trained values ----
        tf, model, variables, y2 = gg.build_reference_model('standin_torch', case, overrides=vals)
        np.testing.assert_array_equal(y, y2)
        obj = model.objective
        grads = tf.gradients(obj, variables)
        g = [np.zeros_like(v) if gi is None else gi.detach().numpy().copy() for v, gi in zip(vals, grads)]
        obj = float(obj)
        assert obj < obj0
        np.testing.assert_allclose(gg.numpy_objective(case, vals), obj, rtol=1e-9)
        o2, g2 = ot.objective_and_gradients(y, dict(zip(NAMES, vals)), s_1=1.0, s_2=1.0, mask_size=case[5])
        np.testing.assert_allclose(o2, obj, rtol=1e-9)
        worst = 0.0
        for k, gi in zip(NAMES, g):
            err = np.abs(g2[k] - gi).max() / max(np.abs(gi).max(), 1e-300)
            worst = max(worst, err)
            assert err <= 1e-6, (k, err)
        np.savez_compressed(os.path.join(gg.OUT, name + '.npz'), y=y, objective=obj, objective_initial=obj0, mask_size=case[5],
                            s_1=1.0, s_2=1.0, adam_iterations=it + 1, adam_lr=lr, kuu_condition=cond, gamma=gamma, guard=guard,
                            **dict(zip(NAMES, vals)), **{'grad_' + k: gi for k, gi in zip(NAMES, g)})
        print('wrote %s: %d Adam iterations, objective %.9f -> %.9f, cond(K_uu) %.3g..%.3g, oracle-vs-reference gradient '
              'deviation %.2e' % (name, it + 1, obj0, obj, cond.min(), cond.max(), worst))


if __name__ == '__main__':
    main()
