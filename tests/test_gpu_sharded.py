"""The product's D-sharded path with TWO ranks (two fresh child processes on the one GPU of the test box, gloo transport):
objective, the eleven gradients and a short Adam run must equal the single-process model's, and a trouble flag raised on one
rank must stop both (the decision is collective).  Multi-GPU reference: SURVEY.md 8(e); the reference itself has
no multi-device code, the single-process model is pinned to it by tests/golden/grad_ref_*.npz."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import golden
from test_gpu_grad import build_model

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURE = 'grad_ref_60_10_15_4_T5'


def _run_ranks(tmp_path, prec, mode, world=2, fixture=FIXTURE):
    port = 29700 + os.getpid() % 1500
    procs, outs = [], []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY='0')
        out = str(tmp_path / ('rank%d_%s_%s.npz' % (rank, prec, mode)))
        outs.append(out)
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, '_sharded_worker.py'), fixture, prec, out, mode],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = []
    for pr in procs:
        try:
            log, _ = pr.communicate(timeout=300)
        except subprocess.TimeoutExpired:          # a rank left waiting in a collective: kill exactly our children
            for q in procs:
                q.kill()
            pytest.fail('a rank hung (collective mismatch between the ranks)')
        logs.append(log.decode(errors='replace'))
    for pr, log in zip(procs, logs):
        assert pr.returncode == 0, log[-3000:]
    return [dict(np.load(o)) for o in outs]


@pytest.mark.parametrize('prec', ['f64', 'mixed'])
def test_two_ranks_equal_one(dev, tmp_path, prec):
    g = golden(FIXTURE)
    single = build_model(g, dev, prec)
    obj1 = float(single.objective)
    terms1 = single.objective_terms.cpu().numpy()
    grads1 = {k: v.cpu().numpy() for k, v in single.gradients().items()}
    single.optimise(5, learning_rate=0.01)
    after1, xu1 = float(single.objective), single.raw['x_u'].cpu().numpy()
    ranks = _run_ranks(tmp_path, prec, 'values')
    d = g['y'].shape[1]
    assert [tuple(r['shard']) for r in ranks] == [(0, d // 2), (d // 2, d)]
    tol = 1e-12 if prec == 'f64' else 1e-9       # same kernels on the same inputs: only the summation order over d differs
    for r in ranks:
        np.testing.assert_allclose(float(r['objective']), obj1, rtol=tol)
        np.testing.assert_allclose(r['terms'], terms1, rtol=tol, atol=tol * np.abs(terms1).max())
        for k, want in grads1.items():
            np.testing.assert_allclose(r['grad_' + k], want, rtol=0, atol=1e3 * tol * max(np.abs(want).max(), 1e-300), err_msg=k)
        assert int(r['descended']) == 1
        # (five Adam steps amplify the last-bit differences of the gradients; in mixed precision the number of n-splits — the
        #  association of the fp32 partial sums — depends on how many output dims a rank holds)
        loose = 1e-7 if prec == 'f64' else 2e-6
        np.testing.assert_allclose(float(r['after']), after1, rtol=loose)
        np.testing.assert_allclose(r['x_u_after'], xu1, rtol=0, atol=loose * np.abs(xu1).max())
    np.testing.assert_array_equal(ranks[0]['x_u_after'], ranks[1]['x_u_after'])      # replicas stay bit-identical


def test_a_flag_on_one_rank_stops_both(dev, tmp_path):
    ranks = _run_ranks(tmp_path, 'mixed', 'flag')
    assert all(int(r['raised']) == 1 for r in ranks)
    assert int(ranks[0]['local_flags']) == 0 and int(ranks[1]['local_flags']) >= 1      # only rank 1 saw the trouble itself


@pytest.mark.parametrize('prec', ['f64', 'mixed'])
def test_over_t_model_two_ranks_equal_one(dev, tmp_path, prec):
    """dp_gp_lvm_t D-sharded over two ranks: the replicated T-atom chain, local V = Psi1^T Y columns, ONE
    scalar exchanged per evaluation and ONE packed gradient all-reduce must give the single-process model's numbers."""
    from dp_gp_lvm_amd.models.dp_gp_lvm import dp_gp_lvm_t
    from test_gpu_model_t import build
    fixture = 'model_t_ref_60_10_15_4_T5'
    g = golden(fixture)
    single = build(dp_gp_lvm_t, g, dev, prec)
    terms1 = single.objective_terms.cpu().numpy()
    grads1 = {k: v.cpu().numpy() for k, v in single.gradients().items()}
    single.optimise(5, learning_rate=0.01)
    after1, xu1 = float(single.objective), single.raw['x_u'].cpu().numpy()
    ranks = _run_ranks(tmp_path, prec, 'values_t', fixture=fixture)
    d = g['y'].shape[1]
    assert [tuple(r['shard']) for r in ranks] == [(0, d // 2), (d // 2, d)]
    tol = 1e-11 if prec == 'f64' else 1e-9       # the same kernels on the same inputs; only the order of the sums over d differs
    for r in ranks:
        np.testing.assert_allclose(r['terms'], terms1, rtol=tol, atol=tol * np.abs(terms1).max())
        np.testing.assert_allclose(r['terms_graph'], terms1, rtol=tol, atol=tol * np.abs(terms1).max())
        # gradients: stage B of this model runs on the matrix pipe (f16-split adjoints) in both precisions, and each rank
        # splits ITS share of the adjoints: agreement at the rounding level of that stage (measured 4e-6 of the largest entry)
        for k, want in grads1.items():
            np.testing.assert_allclose(r['grad_' + k], want, rtol=0, atol=2e-5 * max(np.abs(want).max(), 1e-300), err_msg=k)
        np.testing.assert_allclose(float(r['after']), after1, rtol=2e-6)
        np.testing.assert_allclose(r['x_u_after'], xu1, rtol=0, atol=1e-5 * np.abs(xu1).max())
    np.testing.assert_array_equal(ranks[0]['x_u_after'], ranks[1]['x_u_after'])      # replicas stay bit-identical
