"""
GPU parity tests of the fused per-output ELBO reduction (C ABI dpgp_elbo_fhat) against golden vectors generated from
the reference's own source (reference model built as its unit tests build it, dpgplvm_unitttests.py:24-350; the
SURVEY 8d synthetic problems) and, at the BASELINE.json shapes, against per-d spot values and domain properties.

Tolerances on the per-d f_hat terms and on f_hat (reference fp64):
   prec f64   : rtol 1e-9   (all arithmetic fp64; differs from the reference only by summation order / LAPACK vs ours)
   prec mixed : rtol 2e-5   (psi-statistics fp32 on the matrix cores, Cholesky chain fp64)
   prec f32   : rtol 5e-4   (everything fp32; K_uu jitter 1e-8 is below fp32 resolution)
each with atol = rtol * max|term| of that problem (individual terms can cancel to ~0).
"""
import numpy as np
import pytest
import torch

from conftest import golden
from dp_gp_lvm_amd import ops
from dp_gp_lvm_amd.utils.synthetic import make_problem, CONFIGS

pytestmark = pytest.mark.gpu
RTOL = {'f64': 1e-9, 'mixed': 2e-5, 'f32': 5e-4}


def run(p, dev, prec, algo='auto'):
    t = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float64, device=dev)
    terms, sums, info = ops.elbo_fhat(t(p['y']), t(p['z']), t(p['mu']), t(p['s']), t(p['gamma']), t(p['alpha']),
                                      t(p['beta']), prec=prec, algo=algo)
    torch.cuda.synchronize()
    return terms.cpu().numpy(), sums.cpu().numpy(), info.cpu().numpy()


def check_terms(terms, ref_terms, prec, what):
    rt = RTOL[prec]
    np.testing.assert_allclose(terms, ref_terms, rtol=rt, atol=rt * np.max(np.abs(ref_terms)), err_msg=what)
    np.testing.assert_allclose(terms.sum(), ref_terms.sum(), rtol=rt, err_msg=what + ' f_hat')


@pytest.mark.parametrize('fixture', ['dpgplvm_50_10_25_3_T8', 'dpgplvm_T1_d5', 'dpgplvm_d2', 'plumbing_100_12_20_4',
                                     'script_100_20_25_10'])
@pytest.mark.parametrize('prec', ['f64', 'mixed', 'f32'])
@pytest.mark.parametrize('algo', ['auto', 'plain', 'patch_f16'])
def test_fhat_terms_golden(dev, fixture, prec, algo):
    g = golden(fixture)
    if 'y' not in g:
        g.update(make_problem(int(g['cfg'])))
    terms, sums, info = run(g, dev, prec, algo)
    assert not info.any()
    check_terms(terms, g['fhat_terms'], prec, fixture)
    np.testing.assert_allclose(terms.sum(axis=1), g['fhat_per_d'], rtol=RTOL[prec],
                               atol=RTOL[prec] * np.max(np.abs(g['fhat_terms'])))
    np.testing.assert_allclose(sums[0], terms.sum(), rtol=1e-12)
    np.testing.assert_allclose(sums[1], float(g['kl']), rtol=1e-12)


@pytest.mark.parametrize('cfg,name', [(2, 'spot_C2'), (3, 'spot_C3'), (4, 'spot_C4'), (5, 'spot_C5')])
@pytest.mark.parametrize('prec', ['f64', 'mixed', 'f32'])
def test_fhat_spot_baseline_shapes(dev, cfg, name, prec):
    """Full N, M, Q of BASELINE configs 2/3/4/5 on the 4 output dims the reference was evaluated on (config 4: the reference
    kernel with B = 1 on N-chunks)."""
    import os
    from conftest import GOLDEN
    if not os.path.exists(os.path.join(GOLDEN, name + '.npz')):
        pytest.skip(name + '.npz is not generated yet (oracle/gen_golden.py spot4)')
    g = golden(name)
    p = make_problem(cfg, d_slice=g['dsel'])
    terms, sums, info = run(p, dev, prec)
    assert not info.any()
    check_terms(terms, g['fhat_terms'], prec, name)
    np.testing.assert_allclose(sums[1], float(g['kl']), rtol=1e-12)
    # psi2 checksums through the operator API
    dt = torch.float64 if prec == 'f64' else torch.float32
    t = lambda a: torch.as_tensor(np.asarray(a), dtype=dt, device=dev)
    p2 = ops.psi2(t(p['z']), t(p['mu']), t(p['s']), t(p['gamma']), t(p['alpha'])).double().cpu().numpy()
    rt = 1e-10 if prec == 'f64' else 1e-4
    np.testing.assert_allclose(np.sqrt((p2 * p2).sum(axis=(1, 2))), g['psi_2_fro'], rtol=rt)
    np.testing.assert_allclose(p2[:, g['sample_i'], g['sample_j']], g['psi_2_samples'], rtol=rt,
                               atol=rt * np.abs(g['psi_2_samples']).max())
    # (fp32: the exponent of the pair-tile kernel is a K = 2Q + 1 product of f16-split operands accumulated in fp32, whose
    #  terms are ~10 x the exponent at Q = 20: entries to ~5e-5, relative to the largest row sum here)
    np.testing.assert_allclose(p2.sum(axis=2), g['psi_2_rowsum'], rtol=rt, atol=0 if prec == 'f64' else rt * np.abs(g['psi_2_rowsum']).max())


@pytest.mark.parametrize('cfg', [2, 3, 4, 5])
def test_full_size_properties(dev, cfg):
    """Size-independent properties at the full BASELINE shape (the oracle would take minutes here):
    (1) sharding D commutes: terms of a D-slice equal the slice of the terms (to fp32 summation order); (2) f_hat is the sum of the terms;
    (3) psi2 is additive over a split of the observations and symmetric; (4) mixed vs fp64 agree to the stated tolerance
    on a D-slice; (5) the MFMA and the plain-VALU psi2 agree."""
    p = make_problem(cfg)
    n, d, m, q = CONFIGS[cfg]
    terms, sums, info = run(p, dev, 'mixed')
    assert not info.any() and np.isfinite(terms).all()
    np.testing.assert_allclose(sums[0], terms.sum(), rtol=1e-12)
    sl = np.arange(d // 2, d // 2 + 16)
    ps = make_problem(cfg, d_slice=sl)
    t_sl, _, _ = run(ps, dev, 'mixed')
    # (the number of n-splits depends on how many output dims a GPU holds, so fp32 partial sums associate differently)
    np.testing.assert_allclose(t_sl, terms[sl], rtol=1e-6, atol=1e-6 * np.abs(terms).max())
    t64, _, _ = run(ps, dev, 'f64')
    check_terms(t_sl, t64, 'mixed', 'mixed vs f64')
    t = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float32, device=dev)
    a = ops.psi2(t(ps['z']), t(ps['mu']), t(ps['s']), t(ps['gamma']), t(ps['alpha']))
    h = n // 2 + 3
    a1 = ops.psi2(t(ps['z']), t(ps['mu'][:h]), t(ps['s'][:h]), t(ps['gamma']), t(ps['alpha']))
    a2 = ops.psi2(t(ps['z']), t(ps['mu'][h:]), t(ps['s'][h:]), t(ps['gamma']), t(ps['alpha']))
    torch.testing.assert_close(a1 + a2, a, rtol=1e-5, atol=1e-5 * float(a.max()))
    assert torch.equal(a, a.transpose(1, 2))
    ap = ops.psi2(t(ps['z']), t(ps['mu']), t(ps['s']), t(ps['gamma']), t(ps['alpha']), algo='plain')
    torch.testing.assert_close(ap, a, rtol=1e-4, atol=1e-6 * float(a.max()))


def test_not_positive_definite_is_reported_not_fatal(dev):
    """A negative noise precision for one output makes A = beta T2 + I indefinite for that output only: its info must
    point into the second factorisation (M + j), its Cholesky-dependent terms are NaN, and every other output is intact."""
    p = make_problem(1)
    ref, _, _ = run(p, dev, 'f64')
    bad = 5
    p['beta'] = p['beta'].copy()
    p['beta'][bad] = -500.0
    terms, sums, info = run(p, dev, 'f64')
    m = p['z'].shape[0]
    assert info[bad] > m and (np.delete(info, bad) == 0).all()
    assert np.isnan(terms[bad, [1, 2, 4]]).all()
    np.testing.assert_array_equal(np.delete(terms, bad, axis=0), np.delete(ref, bad, axis=0))


def test_config4_shape_against_c_oracle(dev):
    """BASELINE config 4 shape (N=10000, M=512, Q=20) on 2 of its 256 output dims: exercises the 8x8-patch psi2 grid,
    three K-steps of f16 MFMA, and the global-memory (non-LDS) blocked Cholesky.  No reference-generated golden exists at
    The reference-generated golden at this shape is spot_C4 (test_fhat_spot_baseline_shapes: the reference kernel with B = 1 on
    N-chunks); this test adds two more output dims against the strict build of the C oracle, itself pinned to the reference."""
    from oracle.c_oracle import COracle
    sel = np.array([3, 200])
    p = make_problem(4, d_slice=sel)
    c = COracle(fast=False)                                  # the strict-IEEE build: the checker (c_oracle.py)
    ref, info_ref = c.fhat_terms(p['y'], p['z'], p['mu'], p['s'], p['gamma'], p['alpha'], p['beta'],
                                 nthreads=min(16, c.max_threads))
    assert not info_ref.any()
    for prec in ('mixed', 'f64'):
        terms, sums, info = run(p, dev, prec)
        assert not info.any()
        check_terms(terms, ref, prec, 'config 4 ' + prec)


@pytest.mark.parametrize('shape', [(300, 3, 200, 5), (400, 2, 384, 8), (150, 2, 129, 8)])
@pytest.mark.parametrize('prec', ['f64', 'mixed'])
def test_persistent_chain_beyond_128_inducing_points(dev, shape, prec, monkeypatch):
    """The persistent-workgroup chain (csrc/chain_big.hip: chol(K), chol(K + beta Psi2), L_A = L_K^-1 L_B as a triangular solve
    with a triangular right-hand side) is chosen when D >= 128; forced here on few output dims so that the NumPy oracle stays
    cheap.  M = 200 / 129: identity padding to 256 rows; M = 384: three block columns.  Also against the round-2 path."""
    from oracle import dpgp_oracle as orc
    p = make_problem(shape=shape, seed=31)
    ref = orc.fhat_terms(p['y'], p['z'], p['mu'], p['s'], p['gamma'], p['alpha'], p['beta'])
    monkeypatch.setenv('DPGP_CHAIN_BIG', '1')
    terms, sums, info = run(p, dev, prec)
    assert not info.any()
    check_terms(terms, ref, prec, 'persistent chain %s %s' % (shape, prec))
    np.testing.assert_allclose(sums[0], terms.sum(), rtol=1e-12)
    monkeypatch.setenv('DPGP_CHAIN_BIG', '0')
    old, _, info0 = run(p, dev, prec)
    assert not info0.any()
    np.testing.assert_allclose(terms, old, rtol=1e-9, atol=1e-9 * np.abs(old).max())


def test_persistent_chain_reports_failures_per_output(dev, monkeypatch):
    """as test_not_positive_definite_is_reported_not_fatal, through the persistent chain (M > 128)"""
    monkeypatch.setenv('DPGP_CHAIN_BIG', '1')
    p = make_problem(shape=(200, 4, 160, 3), seed=5)
    ref, _, info = run(p, dev, 'f64')
    assert not info.any()
    p['beta'] = p['beta'].copy()
    p['beta'][2] = -500.0
    terms, sums, info = run(p, dev, 'f64')
    assert info[2] > 160 and (np.delete(info, 2) == 0).all()
    assert np.isnan(terms[2, [1, 2, 4]]).all()
    np.testing.assert_array_equal(np.delete(terms, 2, axis=0), np.delete(ref, 2, axis=0))


def test_sharded_model_single_rank_group(dev):
    """The D-sharded code path of the model object (pack -> all_reduce -> finalize) with a 1-rank process group on the GPU."""
    import os
    import torch.distributed as dist
    from conftest import golden as _g
    from test_gpu_model import build
    g = _g('dpgplvm_50_10_25_3_T8')
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', str(29600 + os.getpid() % 1000))
    created = not dist.is_initialized()
    if created:
        dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
    try:
        model = build(g, dev, 'f64', process_group=dist.group.WORLD)
        np.testing.assert_allclose(float(model.objective), float(g['objective']), rtol=1e-9)
    finally:
        if created:
            dist.destroy_process_group()
