"""
TEST INFRASTRUCTURE, CONTAINER-ONLY: fixtures for the Bayesian GP-LVM wrapper (the B = 1 model of
src/models/gaussian_process.py:132-270, which needs no new kernels).
Run as ``python oracle/gen_golden_bgplvm.py`` in the build container (needs /root/reference; never runs on the GPU box).

Runs the reference's own ``bayesian_gp_lvm(...)`` under the PyTorch stand-in for TensorFlow with steered initial values
(as oracle/gen_golden_grad.py), differentiates its objective with respect to its six trainable variables, in creation order
    gamma_raw [1,Q], alpha_raw [1,1], beta_raw [1,1], x_mean [N,Q], x_u [M,Q], x_var_raw [N,Q]   (gaussian_process.py:172-228)
and checks, before writing: the NumPy stand-in at the same values (1e-11), central differences, and the restatement
oracle/dpgp_oracle_torch.py (the over-T f_hat with one atom and phi = 1: objective 1e-11, gradients 1e-7).  The reference's
own known-answer test for this model (test/unittests/bgplvm_unittests.py TestBGPLVM.test_objective) is among the unit tests
oracle/gen_golden.py runs under the NumPy stand-in.
"""
import importlib
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, REPO)
import numpy as np                                                           # noqa: E402
from oracle import gen_golden_grad as gg                                     # noqa: E402

NAMES = ['gamma_raw', 'alpha_raw', 'beta_raw', 'x_mean', 'x_u', 'x_var_raw']
CASES = {'bgplvm_ref_40_6_12_3': (40, 6, 12, 3, 41), 'bgplvm_ref_70_9_20_4': (70, 9, 20, 4, 42)}   # N, D, M, Q, seed


def build(backend, case, overrides=None):
    for k in [k for k in sys.modules if k == 'tensorflow' or k.startswith('tensorflow.') or k == 'tensorflow_probability'
              or k == 'src' or k.startswith('src.')]:
        del sys.modules[k]
    sys.path[:] = [p for p in sys.path if os.path.basename(p) not in ('standin', 'standin_torch')]
    sys.path[:0] = [os.path.join(HERE, backend), gg.LINK]
    tf = importlib.import_module('tensorflow')
    assert backend in tf.__file__
    gpm = importlib.import_module('src.models.gaussian_process')
    n, d, m, q, seed = case
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
        model = gpm.bayesian_gp_lvm(y_train=y, num_latent_dims=q, num_inducing_points=m)
    finally:
        tf.Variable = real_variable
    variables = tf.get_collection(tf.GraphKeys.TRAINABLE_VARIABLES)
    assert len(variables) == len(NAMES), len(variables)
    return tf, model, variables, y


def numpy_objective(case, values):
    _, model, _, _ = build('standin', case, overrides=values)
    return float(model.objective)


def main():
    from oracle import dpgp_oracle_torch as ot
    for name, case in CASES.items():
        tf, model, variables, y = build('standin_torch', case)
        obj = model.objective
        grads = tf.gradients(obj, variables)
        vals = [v.detach().numpy().copy() for v in variables]
        for nm, v in zip(NAMES, vals):
            assert v.ndim == {'gamma_raw': 2, 'alpha_raw': 2, 'beta_raw': 2, 'x_mean': 2, 'x_u': 2, 'x_var_raw': 2}[nm], (nm, v.shape)
        g = [np.zeros_like(v) if gi is None else gi.detach().numpy().copy() for v, gi in zip(vals, grads)]
        obj = float(obj)
        np.testing.assert_allclose(numpy_objective(case, vals), obj, rtol=1e-11)
        rs = np.random.default_rng(5)
        for _ in range(6):
            dirs = [rs.standard_normal(v.shape) for v in vals]
            h = 1e-5
            fd = (numpy_objective(case, [v + h * e for v, e in zip(vals, dirs)]) -
                  numpy_objective(case, [v - h * e for v, e in zip(vals, dirs)])) / (2 * h)
            an = sum(float(np.sum(gi * e)) for gi, e in zip(g, dirs))
            assert abs(fd - an) <= 2e-6 * max(1.0, abs(an)), (fd, an)
        o2, g2 = ot.objective_bgplvm_and_gradients(y, dict(zip(NAMES, vals)))
        np.testing.assert_allclose(o2, obj, rtol=1e-11)
        for k, gi in zip(NAMES, g):
            np.testing.assert_allclose(g2[k], gi, rtol=1e-7, atol=1e-9 * max(1.0, np.abs(gi).max()), err_msg=k)
        np.savez_compressed(os.path.join(gg.OUT, name + '.npz'), y=y, objective=obj, **dict(zip(NAMES, vals)),
                            **{'grad_' + k: gi for k, gi in zip(NAMES, g)})
        print('wrote %s: objective %.12f' % (name, obj))


if __name__ == '__main__':
    main()
