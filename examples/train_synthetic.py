#!/usr/bin/env python3
"""
Train a DP-GP-LVM on synthetic data with the HIP forward + backward pass — the flow of the reference's
test/synthetic_data_hard_test.py:120-172 (build model, Adam on the objective, log every 100 iterations, save the converged
values under the reference's result keys, src/utils/constants.py:38-72), with `model.optimise` in place of
`tf.train.AdamOptimizer(...).minimize(objective)` inside a `tf.Session`.

    python examples/train_synthetic.py [--n 200] [--d 24] [--m 30] [--q 5] [--t 8] [--iters 500] [--lr 0.01] [--out result.npz]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    for k, v in dict(n=200, d=24, m=30, q=5, t=8, iters=500, seed=1).items():
        ap.add_argument('--' + k, type=int, default=v)
    ap.add_argument('--lr', type=float, default=0.01)
    ap.add_argument('--precision', default='f64', choices=['mixed', 'f64'],
                    help="f64 (default): fp64 forward + streaming backward stage on the matrix pipe, the configuration that follows "
                         "the reference's arithmetic through training; mixed: fp32 psi-statistics, raises when its guard fires")
    ap.add_argument('--out', default='')
    ap.add_argument('--over-t', action='store_true', help='train the over-T formulation dp_gp_lvm_t instead of dp_gp_lvm')
    ap.add_argument('--predict', type=int, default=0, help='hold out this many rows and report their test log-likelihood')
    a = ap.parse_args()
    import torch
    from dp_gp_lvm_amd.models.dp_gp_lvm import dp_gp_lvm, dp_gp_lvm_t

    # three groups of output dims generated from different subsets of a shared latent space (what the DP should recover)
    rng = np.random.default_rng(a.seed)
    np.random.seed(a.seed)
    x = rng.standard_normal((a.n, 3))
    groups = np.array_split(np.arange(a.d), 3)
    y = np.empty((a.n, a.d))
    for gi, idx in enumerate(groups):
        w = rng.standard_normal((2, len(idx)))
        y[:, idx] = np.tanh(x[:, [gi, (gi + 1) % 3]]) @ w + 0.05 * rng.standard_normal((a.n, len(idx)))
    y = (y - y.mean(axis=0)) / y.std(axis=0)

    y_all = y
    if a.predict:
        y, y_held_out = y_all[:-a.predict], y_all[-a.predict:]
    factory = dp_gp_lvm_t if a.over_t else dp_gp_lvm
    model = factory(y_train=y, num_inducing_points=a.m, num_latent_dims=a.q, truncation_level=a.t,
                    device=torch.device('cuda', 0), precision=a.precision,
                    **({} if a.over_t else {'backward_precision': 'mixed'}))
    print('Training DP-GP-LVM: N=%d D=%d M=%d Q=%d T=%d, %d Adam iterations, lr %g' % (a.n, a.d, a.m, a.q, a.t, a.iters, a.lr))
    t0 = time.time()

    def log(c):
        if c % 100 == 0:
            print('  GP-DP opt iter {:5}: {}'.format(c, float(model.objective)))
    model.optimise(a.iters, learning_rate=a.lr, callback=log)
    train_opt_time = time.time() - t0
    print('Final iter {:5}:\n  GP-DP: {}\nTime to optimise: {} s'.format(a.iters - 1, float(model.objective), train_opt_time))
    if a.predict and not a.over_t:
        lower_bound, x_test_mean, x_test_covar, test_ll = model.predict_new_latent_variables(y_held_out)
        print('held-out rows: prediction lower bound {}, test log-likelihood {} (at the nearest-neighbour q(X*))'.format(
            float(lower_bound), float(test_ll)))
    phi = model.assignments.cpu().numpy()
    print('group assignment of the output dims (argmax of q(Z)):', phi.argmax(axis=1))
    if a.out:
        # the reference's result schema (src/utils/constants.py:38-72): every ResultKeys array, under its key string
        from dp_gp_lvm_amd.utils.results import save_results
        test = {}
        if a.predict and not a.over_t:
            test = dict(y_test=y_held_out, x_test_mean=x_test_mean, x_test_covar=x_test_covar)
        save_results(a.out, model, y, extra=dict(train_opt_time=train_opt_time), **test)
        print('saved', a.out)


if __name__ == '__main__':
    main()
