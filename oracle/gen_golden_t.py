"""
TEST INFRASTRUCTURE, CONTAINER-ONLY: fixtures for the over-T formulation ``dp_gp_lvm_t``.
Run as ``python oracle/gen_golden_t.py`` in the build container (needs /root/reference; never runs on the GPU box).

Runs the reference's own, unmodified ``dp_gp_lvm_t(...)`` constructor (src/models/dp_gp_lvm.py:513-676) under the two
stand-ins for TensorFlow, exactly as oracle/gen_golden_grad.py does for ``dp_gp_lvm`` (same steering of the initial VALUES
of the eleven trainable variables, same three checks: NumPy stand-in at the same values, central differences, the
restatement oracle/dpgp_oracle_torch.py:objective_t), plus the reference's own known answer for this model
(test/unittests/dpgplvm_unitttests.py:544-548): at the un-steered initialisation, where all atoms are equal, the over-T and
the over-D objectives coincide.  Fixtures are data only: y, the raw variable values, the objective and its gradients.
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

OUT = gg.OUT
NAMES = gg.NAMES
CASES = {   # name: (N, D, M, Q, T, mask_size, seed)
    'model_t_ref_40_6_12_3_T4': (40, 6, 12, 3, 4, 1, 21),
    'model_t_ref_60_10_15_4_T5': (60, 10, 15, 4, 5, 1, 22),
}


def build(backend, case, overrides=None, steer=True, factory='dp_gp_lvm_t'):
    for k in [k for k in sys.modules if k == 'tensorflow' or k.startswith('tensorflow.') or k == 'tensorflow_probability'
              or k == 'src' or k.startswith('src.')]:
        del sys.modules[k]
    sys.path[:] = [p for p in sys.path if os.path.basename(p) not in ('standin', 'standin_torch')]
    sys.path[:0] = [os.path.join(HERE, backend), gg.LINK]
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
        if trainable and (steer or it is not None):
            init = np.asarray(initial_value, dtype=np.float64)
            initial_value = next(it) if it is not None else init + 0.25 * pert.standard_normal(init.shape)
        return real_variable(initial_value, dtype=dtype, trainable=trainable, **kw)
    tf.Variable = steered_variable
    try:
        kw = dict(y_train=y, num_latent_dims=q, num_inducing_points=m, truncation_level=t,
                  alpha_prior_params=np.array([1.0, 1.0]), mask_size=mask)
        if factory == 'dp_gp_lvm_t':
            kw['seed'] = seed
        model = getattr(dgl, factory)(**kw)
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
        # the reference's known answer: equal atoms => both formulations give the same objective (values handed over, since
        # the two constructors draw their random initial values in a different order)
        _, model_t0, vars_t0, _ = build('standin', case, steer=False)
        vals0 = [np.asarray(v).copy() for v in vars_t0]
        _, model_d0, _, _ = build('standin', case, overrides=vals0, factory='dp_gp_lvm')
        np.testing.assert_allclose(float(model_t0.objective), float(model_d0.objective), rtol=1e-9)
        tf, model, variables, y = build('standin_torch', case)
        obj = model.objective
        grads = tf.gradients(obj, variables)
        vals = [v.detach().numpy().copy() for v in variables]
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
        n, d, m, q, t, mask, seed = case
        o2, g2 = ot.objective_t_and_gradients(y, dict(zip(NAMES, vals)), s_1=1.0, s_2=1.0, mask_size=mask)
        np.testing.assert_allclose(o2, obj, rtol=1e-11)
        for k, gi in zip(NAMES, g):
            np.testing.assert_allclose(g2[k], gi, rtol=1e-7, atol=1e-9 * max(1.0, np.abs(gi).max()), err_msg=k)
        # the over-D objective at the same (steered) values, for the record: the two models differ away from equal atoms
        _, model_d, _, _ = build('standin', case, overrides=vals, factory='dp_gp_lvm')
        np.savez_compressed(os.path.join(OUT, name + '.npz'), y=y, objective=obj, objective_over_d=float(model_d.objective),
                            objective_init=float(model_t0.objective), mask_size=mask, s_1=1.0, s_2=1.0,
                            **dict(zip(NAMES, vals)), **{'init_' + k: v for k, v in zip(NAMES, vals0)},
                            **{'grad_' + k: gi for k, gi in zip(NAMES, g)})
        print('wrote %s: objective %.12f (over-D model at the same values %.12f; at the equal-atoms initialisation both %.12f)'
              % (name, obj, float(model_d.objective), float(model_t0.objective)))


if __name__ == '__main__':
    main()
