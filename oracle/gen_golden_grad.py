"""
TEST INFRASTRUCTURE, CONTAINER-ONLY: gradient fixtures for the backward pass of the fused ELBO.
Run as ``python oracle/gen_golden_grad.py`` in the build container (needs /root/reference; never runs on the GPU box).

The reference trains by Adam on ``tf.gradients(objective)`` (test/synthetic_data_hard_test.py:143-155).  This script runs
the reference's own, unmodified ``dp_gp_lvm(...)`` constructor (src/models/dp_gp_lvm.py:21-154) under
``oracle/standin_torch`` — an eager PyTorch fp64 module named ``tensorflow`` — so that the objective it builds carries an
autograd graph, and differentiates it with respect to the reference's eleven trainable variables, in creation order:
    x_mean [N,Q], x_var_raw [N,Q], x_u [M,Q]                       (dp_gp_lvm.py:63-74)
    dp logits [D/mask,T], gamma_1_raw [T-1], gamma_2_raw [T-1], w_1_raw, w_2_raw   (dirichlet_process.py:40-59)
    gamma_atoms_raw [T,Q], alpha_atoms_raw [T,1], beta_atoms_raw [T,1]            (dp_gp_lvm.py:84-94)
Only initial VALUES are steered: ``numpy.random`` is seeded, and the stand-in's own ``tf.Variable`` adds 0.25 N(0,1) to
the initial value of every trainable variable (so that the point is not the reference's degenerate all-atoms-equal,
all-variances-one initialisation); every line that computes the objective is the reference's.
Checks before a fixture is written:
  * the same constructor under the NumPy stand-in, at the same variable values, gives the same objective (two
    independent stand-ins);
  * central finite differences of that NumPy objective along 6 random directions agree with the autograd gradient;
  * ``oracle/dpgp_oracle_torch.py`` (the restatement that will check the HIP backward pass) reproduces objective and
    gradients.
Fixtures are data only: y, the raw variable values, the objective and its gradients.
"""

import importlib
import os
import subprocess
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
REF = '/root/reference'
LINK_DIR = '/tmp/dpgp_reflink'
os.makedirs(LINK_DIR, exist_ok=True)
LINK = os.path.join(LINK_DIR, 'dp_gp_lvm')
if not os.path.islink(LINK):
    os.symlink(REF, LINK)

import numpy as np                                                           # noqa: E402

OUT = os.path.join(REPO, 'tests', 'golden')
NAMES = ['x_mean', 'x_var_raw', 'x_u', 'dp_logits', 'gamma_1_raw', 'gamma_2_raw', 'w_1_raw', 'w_2_raw',
         'gamma_atoms_raw', 'alpha_atoms_raw', 'beta_atoms_raw']
CASES = {   # name: (N, D, M, Q, T, mask_size, seed)
    'grad_ref_40_6_12_3_T4': (40, 6, 12, 3, 4, 1, 11),
    'grad_ref_60_10_15_4_T5': (60, 10, 15, 4, 5, 1, 12),      # (mask_size > 1 hits the removed alias np.int in the reference)
}


def build_reference_model(backend, case, overrides=None):
    """Run the reference constructor under the stand-in `backend` ('standin' = NumPy, 'standin_torch' = PyTorch).
    overrides: optional list of 11 arrays that replace the initial values (used for the finite differences)."""
    for k in [k for k in sys.modules if k == 'tensorflow' or k.startswith('tensorflow.') or k == 'tensorflow_probability'
              or k == 'src' or k.startswith('src.')]:
        del sys.modules[k]
    sys.path[:] = [p for p in sys.path if os.path.basename(p) not in ('standin', 'standin_torch')]
    sys.path[:0] = [os.path.join(HERE, backend), LINK]
    tf = importlib.import_module('tensorflow')
    assert backend in tf.__file__
    dgl = importlib.import_module('src.models.dp_gp_lvm')
    n, d, m, q, t, mask, seed = case
    rng = np.random.default_rng(seed)
    y = rng.standard_normal((n, d)) + 0.7 * np.outer(rng.standard_normal(n), rng.standard_normal(d))
    y = (y - y.mean(axis=0)) / y.std(axis=0)
    pert = np.random.default_rng(seed + 1000)
    tf.reset_default_graph()
    np.random.seed(seed)
    it = iter(overrides) if overrides is not None else None
    real_variable = tf.Variable

    def steered_variable(initial_value=None, dtype=None, trainable=True, **kw):
        if trainable:
            init = np.asarray(initial_value, dtype=np.float64)
            initial_value = next(it) if it is not None else init + 0.25 * pert.standard_normal(init.shape)
        return real_variable(initial_value, dtype=dtype, trainable=trainable, **kw)
    tf.Variable = steered_variable
    try:
        model = dgl.dp_gp_lvm(y_train=y, num_latent_dims=q, num_inducing_points=m, truncation_level=t,
                              alpha_prior_params=np.array([1.0, 1.0]), mask_size=mask)
    finally:
        tf.Variable = real_variable
    variables = tf.get_collection(tf.GraphKeys.TRAINABLE_VARIABLES)
    assert len(variables) == len(NAMES), len(variables)
    return tf, model, variables, y


def numpy_objective(case, values):
    _, model, _, _ = build_reference_model('standin', case, overrides=values)
    return float(model.objective)


def main():
    from oracle import dpgp_oracle_torch as ot
    for name, case in CASES.items():
        tf, model, variables, y = build_reference_model('standin_torch', case)
        obj = model.objective
        grads = tf.gradients(obj, variables)
        vals = [v.detach().numpy().copy() for v in variables]
        g = [np.zeros_like(v) if gi is None else gi.detach().numpy().copy() for v, gi in zip(vals, grads)]
        obj = float(obj)
        # (1) NumPy stand-in, same constructor, at the same variable values (the PCA initialisation of the two stand-ins
        #     differs by eigenvector signs, so the values are handed over rather than re-drawn)
        np.testing.assert_allclose(numpy_objective(case, vals), obj, rtol=1e-11)
        # (2) central differences of the NumPy objective along random directions
        rs = np.random.default_rng(5)
        for _ in range(6):
            dirs = [rs.standard_normal(v.shape) for v in vals]
            h = 1e-5
            fp = numpy_objective(case, [v + h * e for v, e in zip(vals, dirs)])
            fm = numpy_objective(case, [v - h * e for v, e in zip(vals, dirs)])
            fd = (fp - fm) / (2 * h)
            an = sum(float(np.sum(gi * e)) for gi, e in zip(g, dirs))
            assert abs(fd - an) <= 2e-6 * max(1.0, abs(an)), (fd, an)
        # (3) the oracle restatement
        n, d, m, q, t, mask, seed = case
        o2, g2 = ot.objective_and_gradients(y, dict(zip(NAMES, vals)), s_1=1.0, s_2=1.0, mask_size=mask)
        np.testing.assert_allclose(o2, obj, rtol=1e-11)
        for k, gi in zip(NAMES, g):
            np.testing.assert_allclose(g2[k], gi, rtol=1e-7, atol=1e-9 * max(1.0, np.abs(gi).max()), err_msg=k)
        np.savez_compressed(os.path.join(OUT, name + '.npz'), y=y, objective=obj, mask_size=mask, s_1=1.0, s_2=1.0,
                            **dict(zip(NAMES, vals)), **{'grad_' + k: gi for k, gi in zip(NAMES, g)})
        print('wrote %s: objective %.12f, |grad| per variable: %s' %
              (name, obj, ' '.join('%s %.3g' % (k, np.linalg.norm(gi)) for k, gi in zip(NAMES, g))))


if __name__ == '__main__':
    sys.path.insert(0, REPO)
    main()
