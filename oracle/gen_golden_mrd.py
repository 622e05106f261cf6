"""
TEST INFRASTRUCTURE, CONTAINER-ONLY: fixtures for the manifold-relevance-determination wrapper (SURVEY.md 8f, row 4:
src/models/gaussian_process.py:551-664, V views with one B = 1 kernel and one set of inducing inputs each, a shared q(X)).
Run as ``python oracle/gen_golden_mrd.py`` in the build container (needs /root/reference; never runs on the GPU box).

Runs the reference's own ``manifold_relevance_determination(...)`` under the PyTorch stand-in for TensorFlow with steered
initial values (as oracle/gen_golden_grad.py) and differentiates its objective with respect to its trainable variables, in
creation order (gaussian_process.py:577-606)
    gamma_raw_v [1,Q] (all views), alpha_raw_v [1,1], beta_raw_v [1,1], x_mean [N,Q], x_var_raw [N,Q], x_u_v [M,Q]
Checks before writing: the NumPy stand-in at the same values (1e-11), central differences, and the restatement
oracle/dpgp_oracle_torch.py:objective_mrd (objective 1e-11, gradients 1e-7).
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

CASES = {'mrd_ref_50_2views_12_3': (50, (5, 7), 12, 3, 51), 'mrd_ref_60_3views_15_4': (60, (4, 6, 5), 15, 4, 52)}   # N, Ds, M, Q, seed


def build(backend, case, overrides=None):
    for k in [k for k in sys.modules if k == 'tensorflow' or k.startswith('tensorflow.') or k == 'tensorflow_probability'
              or k == 'src' or k.startswith('src.')]:
        del sys.modules[k]
    sys.path[:] = [p for p in sys.path if os.path.basename(p) not in ('standin', 'standin_torch')]
    sys.path[:0] = [os.path.join(HERE, backend), gg.LINK]
    tf = importlib.import_module('tensorflow')
    assert backend in tf.__file__
    gpm = importlib.import_module('src.models.gaussian_process')
    n, dims, m, q, seed = case
    rng = np.random.default_rng(seed)
    shared = rng.standard_normal((n, 2))
    views = []
    for d in dims:
        y = np.tanh(shared) @ rng.standard_normal((2, d)) + 0.5 * rng.standard_normal((n, d))
        views.append((y - y.mean(axis=0)) / y.std(axis=0))
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
        model = gpm.manifold_relevance_determination(views_train=views, num_latent_dims=q, num_inducing_points=m)
    finally:
        tf.Variable = real_variable
    variables = tf.get_collection(tf.GraphKeys.TRAINABLE_VARIABLES)
    return tf, model, variables, views


def numpy_objective(case, values):
    _, model, _, _ = build('standin', case, overrides=values)
    return float(model.objective)


def main():
    from oracle import dpgp_oracle_torch as ot
    for name, case in CASES.items():
        tf, model, variables, views = build('standin_torch', case)
        names = ot.mrd_names(len(views))
        assert len(variables) == len(names), (len(variables), len(names))
        obj = model.objective
        grads = tf.gradients(obj, variables)
        vals = [v.detach().numpy().copy() for v in variables]
        n, dims, m, q, _ = case
        for nm, v in zip(names, vals):
            want = (1, q) if nm.startswith('gamma') else (1, 1) if nm[:4] in ('alph', 'beta') else (m, q) if nm.startswith('x_u') else (n, q)
            assert v.shape == want, (nm, v.shape, want)
        g = [np.zeros_like(v) if gi is None else gi.detach().numpy().copy() for v, gi in zip(vals, grads)]
        obj = float(obj)
        np.testing.assert_allclose(numpy_objective(case, vals), obj, rtol=1e-11)
        rs = np.random.default_rng(5)
        for _ in range(4):
            dirs = [rs.standard_normal(v.shape) for v in vals]
            h = 1e-5
            fd = (numpy_objective(case, [v + h * e for v, e in zip(vals, dirs)]) -
                  numpy_objective(case, [v - h * e for v, e in zip(vals, dirs)])) / (2 * h)
            an = sum(float(np.sum(gi * e)) for gi, e in zip(g, dirs))
            assert abs(fd - an) <= 2e-6 * max(1.0, abs(an)), (fd, an)
        o2, g2 = ot.objective_mrd_and_gradients(views, dict(zip(names, vals)))
        np.testing.assert_allclose(o2, obj, rtol=1e-11)
        for k, gi in zip(names, g):
            np.testing.assert_allclose(g2[k], gi, rtol=1e-7, atol=1e-9 * max(1.0, np.abs(gi).max()), err_msg=k)
        np.savez_compressed(os.path.join(gg.OUT, name + '.npz'), objective=obj, num_views=len(views),
                            **{'view_%d' % i: y for i, y in enumerate(views)}, **dict(zip(names, vals)),
                            **{'grad_' + k: gi for k, gi in zip(names, g)})
        print('wrote %s: objective %.12f' % (name, obj))


if __name__ == '__main__':
    main()
