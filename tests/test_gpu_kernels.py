"""
GPU parity tests of the kernel operators (through the C ABI, via dp_gp_lvm_amd.ops) against the golden vectors the
reference's own source produced (tests/golden/kernel_b1.npz = TestRbfKernel inputs, kernel_b7.npz = TestRbfBatchKernel;
/root/reference/test/unittests/kernel_unittests.py:150-829) and against the CPU oracle on seeded inputs.

Tolerances (stated per north star, fp64 reference -> fp32 kernels):
  fp64 kernels : rtol 1e-10 on every entry (atol 1e-13 * max|ref| for entries that underflow relative precision)
  fp32 kernels : rtol 2e-5 on gram / psi1 entries, 1e-4 on psi2 entries (atol 1e-6 * max|ref|)
"""
import numpy as np
import pytest
import torch

from conftest import golden
from dp_gp_lvm_amd import ops
from oracle import dpgp_oracle as orc

pytestmark = pytest.mark.gpu

TOL = {torch.float64: dict(rtol=1e-10, atol_rel=1e-13), torch.float32: dict(rtol=2e-5, atol_rel=1e-6)}
TOL_PSI2 = {torch.float64: dict(rtol=1e-10, atol_rel=1e-13), torch.float32: dict(rtol=1e-4, atol_rel=1e-6)}


def close(got, ref, tol, what=''):
    got = got.detach().cpu().numpy().astype(np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    np.testing.assert_allclose(got, ref, rtol=tol['rtol'], atol=tol['atol_rel'] * np.max(np.abs(ref)), err_msg=what)


def T(a, dt, dev):
    return torch.as_tensor(np.asarray(a), dtype=dt, device=dev)


@pytest.mark.parametrize('fixture', ['kernel_b1', 'kernel_b7'])
@pytest.mark.parametrize('dt', [torch.float64, torch.float32])
def test_covariance_matrix_all_variants(dev, fixture, dt):
    g = golden(fixture)
    gam, al, be = T(g['gamma'], dt, dev), T(g['alpha'], dt, dev), T(g['beta'], dt, dev)
    combos = {'xx': ('x0', None), 'x01': ('x0', 'x1'), 'x10': ('x1', 'x0'), 'uu_same': ('x_u', 'x_u'),
              'uu': ('x_u', None)}
    for tag, (a, c) in combos.items():
        for noise in (0, 1):
            for jit in (0, 1):
                out = ops.ard_rbf_gram(T(g[a], dt, dev), None if c is None else T(g[c], dt, dev), gam, al, be,
                                       include_noise=bool(noise), include_jitter=bool(jit))
                close(out, g['gram_%s_n%d_j%d' % (tag, noise, jit)], TOL[dt], 'gram %s n%d j%d' % (tag, noise, jit))


@pytest.mark.parametrize('fixture', ['kernel_b1', 'kernel_b7'])
@pytest.mark.parametrize('dt', [torch.float64, torch.float32])
def test_diag_psi0_psi1(dev, fixture, dt):
    g = golden(fixture)
    gam, al, be = T(g['gamma'], dt, dev), T(g['alpha'], dt, dev), T(g['beta'], dt, dev)
    n = g['x0'].shape[0]
    for noise in (0, 1):
        for jit in (0, 1):
            close(ops.ard_rbf_diag(n, al, be, bool(noise), bool(jit)), g['diag_n%d_j%d' % (noise, jit)], TOL[dt], 'diag')
    close(ops.psi0(g['x_mean'].shape[0], al), g['psi_0'], TOL[dt], 'psi0')
    p1 = ops.psi1(T(g['x_u'], dt, dev), T(g['x_mean'], dt, dev), T(g['x_var'], dt, dev), gam, al)
    close(p1, g['psi_1'], TOL[dt], 'psi1')


@pytest.mark.parametrize('fixture', ['kernel_b1', 'kernel_b7'])
@pytest.mark.parametrize('dt', [torch.float64, torch.float32])
@pytest.mark.parametrize('algo', ['auto', 'plain', 'mfma_f32'])
def test_psi2_golden(dev, fixture, dt, algo):
    g = golden(fixture)
    p2 = ops.psi2(T(g['x_u'], dt, dev), T(g['x_mean'], dt, dev), T(g['x_var'], dt, dev), T(g['gamma'], dt, dev),
                  T(g['alpha'], dt, dev), algo=algo)
    close(p2, g['psi_2'], TOL_PSI2[dt], 'psi2 ' + algo)
    assert torch.equal(p2, p2.transpose(1, 2)), 'psi2 must be exactly symmetric'


@pytest.mark.parametrize('shape', [(3, 37, 5, 1), (2, 300, 33, 3), (5, 257, 64, 10), (2, 500, 100, 20), (1, 130, 130, 30),
                                   (3, 1000, 128, 10)])
@pytest.mark.parametrize('dt', [torch.float64, torch.float32])
def test_psi_statistics_vs_oracle_ragged(dev, shape, dt):
    """Ragged shapes (M, N not multiples of the tiles; Q from 1 to the maximum 30) against the NumPy oracle."""
    b, n, m, q = shape
    rng = np.random.default_rng(sum(shape))
    z, mu = rng.standard_normal((m, q)), rng.standard_normal((n, q))
    s = np.exp(0.5 * rng.standard_normal((n, q)))
    gam, al = np.exp(0.3 * rng.standard_normal((b, q))), np.exp(0.3 * rng.standard_normal((b, 1)))
    y = rng.standard_normal((n, b))
    args = [T(a, dt, dev) for a in (z, mu, s, gam, al)]
    ref2 = orc.psi2(z, mu, s, gam, al)
    close(ops.psi2(*args), ref2, TOL_PSI2[dt], 'psi2')
    close(ops.psi2(*args, algo='mfma_f32'), ref2, TOL_PSI2[dt], 'psi2 fp32-MFMA variant')
    if dt == torch.float32:
        close(ops.psi2(*args, algo='patch_f16'), ref2, TOL_PSI2[dt], 'psi2 per-observation patch kernel')
    close(ops.psi1(*args), orc.psi1(z, mu, s, gam, al), TOL[dt], 'psi1')
    tol = dict(TOL[dt])
    if dt == torch.float32:
        tol['atol_rel'] = 2e-5      # a signed sum over n of fp32 terms
    close(ops.psi1T_y(*args, T(y, dt, dev)), orc.psi1T_y(z, mu, s, gam, al, y), tol, 'psi1T_y')


@pytest.mark.parametrize('shape', [(2, 1300, 100, 20), (3, 700, 70, 7), (1, 2500, 200, 12)])
def test_psi2_eight_wave_workgroups(dev, shape, monkeypatch):
    """The pair-tile kernel with eight waves on the whole LDS of a compute unit (psi2_pairs.hip, NW = 8: chosen for >= 512 pair
    tiles when the observations of a workgroup need several chunks — config 4), forced here on shapes the oracle can follow:
    several chunks per workgroup, ragged last chunk, tile counts that are not multiples of 8."""
    b, n, m, q = shape
    rng = np.random.default_rng(sum(shape))
    z, mu = rng.standard_normal((m, q)), rng.standard_normal((n, q))
    s = np.exp(0.5 * rng.standard_normal((n, q)))
    gam, al = np.exp(0.3 * rng.standard_normal((b, q))), np.exp(0.3 * rng.standard_normal((b, 1)))
    args = [T(a, torch.float32, dev) for a in (z, mu, s, gam, al)]
    ref2 = orc.psi2(z, mu, s, gam, al)
    monkeypatch.setenv('DPGP_PP_NW', '8')
    monkeypatch.setenv('DPGP_PSI2_NS', '1')
    p8 = ops.psi2(*args)
    close(p8, ref2, TOL_PSI2[torch.float32], 'psi2, eight waves')
    monkeypatch.setenv('DPGP_PP_NW', '4')
    p4 = ops.psi2(*args)
    torch.testing.assert_close(p8, p4, rtol=1e-5, atol=1e-5 * float(p4.max()))


@pytest.mark.parametrize('dt', [torch.float64, torch.float32])
def test_psi2_far_from_origin_is_translation_invariant(dev, dt):
    """q(X) and Z far from the origin: the kernel centres its coordinates, so fp32 accuracy must not degrade."""
    rng = np.random.default_rng(5)
    b, n, m, q = 2, 400, 48, 6
    z, mu = rng.standard_normal((m, q)), rng.standard_normal((n, q))
    s = np.exp(0.5 * rng.standard_normal((n, q)))
    gam, al = np.exp(0.3 * rng.standard_normal((b, q))), np.ones((b, 1))
    ref = orc.psi2(z, mu, s, gam, al)
    shift = 50.0
    got = ops.psi2(T(z + shift, dt, dev), T(mu + shift, dt, dev), T(s, dt, dev), T(gam, dt, dev), T(al, dt, dev))
    tol = dict(TOL_PSI2[dt])
    if dt == torch.float32:
        tol['rtol'] = 2e-3      # the inputs themselves lose 3 digits when stored as fp32 at offset 50
    close(got, ref, tol, 'shifted psi2')


def test_psi_statistics_outside_the_f16_range_are_poisoned_not_wrong(dev):
    """The default fp32 psi kernels split their operands into f16 pairs: the per-(n, m) log-terms must stay below ~3e4, i.e.
    |z - mean z| and |mu - mean z| within ~90 length scales.  Beyond that the clamp is DETECTED and the result is NaN (the fused
    ELBO then reports a failed factorisation) — never a silently wrong number; the exact-fp32 kernel (algo='mfma_f32') and
    fp64 have no such limit and still match the oracle.  Inducing points at +-150 length scales around observations at 0."""
    rng = np.random.default_rng(9)
    b, n, m, q = 2, 300, 40, 3
    z = rng.standard_normal((m, q))
    z[: m // 2] += 150.0
    z[m // 2:] -= 150.0                                              # column means ~0, |z - c| ~ 150
    mu = rng.standard_normal((n, q)) * 0.5
    mu[:50] += 150.0                                                 # some observations sit at the far cluster
    s = np.exp(0.3 * rng.standard_normal((n, q)))
    gam, al = np.ones((b, q)), np.ones((b, 1))
    ref = orc.psi2(z, mu, s, gam, al)
    f32, f64 = torch.float32, torch.float64
    got = ops.psi2(T(z, f32, dev), T(mu, f32, dev), T(s, f32, dev), T(gam, f32, dev), T(al, f32, dev))
    assert bool(torch.isnan(got).any()), 'out-of-range operands must poison the f16-split result'
    exact = ops.psi2(T(z, f32, dev), T(mu, f32, dev), T(s, f32, dev), T(gam, f32, dev), T(al, f32, dev), algo='mfma_f32')
    close(exact, ref, dict(rtol=2e-3, atol_rel=1e-6), 'exact-fp32 psi2 far outside the f16 range')
    dbl = ops.psi2(T(z, f64, dev), T(mu, f64, dev), T(s, f64, dev), T(gam, f64, dev), T(al, f64, dev))
    close(dbl, ref, TOL_PSI2[f64], 'fp64 psi2 far outside the f16 range')
    # the fused ELBO in mixed precision reports it (NaN terms + info != 0); fp64 evaluates
    y = rng.standard_normal((n, b))
    be = np.ones((b, 1))
    args = [T(a, f64, dev) for a in (y, z, mu, s, gam, al, be)]
    terms, sums, info = ops.elbo_fhat(*args, prec='mixed')
    assert int((info != 0).sum()) == b and bool(torch.isnan(terms).any())
    terms64, _, info64 = ops.elbo_fhat(*args, prec='f64')
    assert int(info64.abs().max()) == 0
    want = orc.fhat_terms(y, z, mu, s, gam, al, be)
    np.testing.assert_allclose(terms64.cpu().numpy(), want, rtol=1e-8, atol=1e-8 * np.abs(want).max())


def test_fp64_psi_kernels_propagate_nan(dev):
    """NaN in gives NaN out for the fp64 streaming kernels (their table-based exp2 once clamped a NaN argument to 2^-1100 = 0;
    ADVICE r2): the reference (TensorFlow) propagates NaN."""
    g = golden('kernel_b7')
    mu = np.array(g['x_mean'], dtype=np.float64, copy=True)
    mu[3, 1] = np.nan
    args = [T(a, torch.float64, dev) for a in (g['x_u'], mu, g['x_var'], g['gamma'], g['alpha'])]
    p2 = ops.psi2(*args).cpu().numpy()
    assert np.isnan(p2).any(), 'psi2 (fp64) swallowed a NaN latent mean'


@pytest.mark.parametrize('m', [1, 7, 16, 20, 25, 64, 100, 128, 130])
@pytest.mark.parametrize('dt', [torch.float64, torch.float32])
@pytest.mark.parametrize('algo', ['auto', 'plain'])
def test_potrf_trsm(dev, m, dt, algo):
    rng = np.random.default_rng(m)
    b, k = 3, 9
    a = rng.standard_normal((b, m, m + 3))
    a = a @ a.transpose(0, 2, 1) + 0.5 * m * np.eye(m)
    rhs = rng.standard_normal((b, m, k))
    l_ref = np.linalg.cholesky(a)
    tol = dict(rtol=1e-11, atol_rel=1e-13) if dt == torch.float64 else dict(rtol=2e-4, atol_rel=2e-6)
    l, info = ops.potrf_batched(T(a, dt, dev), algo=algo)
    assert int(info.abs().max()) == 0
    close(l, l_ref, tol, 'potrf')
    x = ops.trsm_batched(T(l_ref, dt, dev), T(rhs, dt, dev), algo=algo)
    x_ref = np.stack([np.linalg.solve(l_ref[i], rhs[i]) for i in range(b)])
    close(x, x_ref, tol, 'trsm')


@pytest.mark.parametrize('shape', [(2, 40, 65), (3, 100, 150), (1, 130, 512), (2, 16, 64)])
def test_trsm_many_right_hand_sides(dev, shape):
    """More than 64 right-hand sides: one workgroup per (matrix, chunk of 64 columns)."""
    b, m, k = shape
    rng = np.random.default_rng(m + k)
    a = rng.standard_normal((b, m, m + 3))
    a = a @ a.transpose(0, 2, 1) + 0.5 * m * np.eye(m)
    l_ref = np.linalg.cholesky(a)
    rhs = rng.standard_normal((b, m, k))
    x = ops.trsm_batched(T(l_ref, torch.float64, dev), T(rhs, torch.float64, dev))
    x_ref = np.stack([np.linalg.solve(l_ref[i], rhs[i]) for i in range(b)])
    close(x, x_ref, dict(rtol=1e-10, atol_rel=1e-12), 'trsm')


@pytest.mark.parametrize('n', [7, 5000, 300001])
def test_trouble_flag(dev, n):
    """dpgp_trouble_flag: 1.0 iff any info != 0 or any non-finite value among flat[0..n) (the value optimise() branches on where the
    reference's tf.cholesky raises); the flag word itself lies behind the scanned range."""
    from dp_gp_lvm_amd import _lib
    l = _lib.lib()
    st = torch.cuda.current_stream().cuda_stream
    rng = np.random.default_rng(n)
    flat = torch.as_tensor(np.append(rng.standard_normal(n), 123.0), dtype=torch.float64, device=dev)
    info = torch.zeros(37, dtype=torch.int32, device=dev)

    def flag():
        _lib.check(l.dpgp_trouble_flag(n, flat.data_ptr(), info.numel(), info.data_ptr(), flat[-1:].data_ptr(), st), 'dpgp_trouble_flag')
        return float(flat[-1])
    assert flag() == 0.0
    for bad in (float('nan'), float('inf'), -float('inf')):
        pos = int(rng.integers(0, n))
        keep = float(flat[pos])
        flat[pos] = bad
        assert flag() == 1.0
        flat[pos] = keep
        assert flag() == 0.0
    info[int(rng.integers(0, 37))] = -2
    assert flag() == 1.0


@pytest.mark.parametrize('m', [128, 256, 512, 200])
def test_inverse_of_a_lower_triangular_factor(dev, m):
    """tf.matrix_triangular_solve(l, eye): the persistent-workgroup solve with every block row stored (M a multiple of 128), the
    column-chunk kernel otherwise; lower, exact zeros above the diagonal."""
    rng = np.random.default_rng(m + 1)
    b = 3
    a = rng.standard_normal((b, m, m + 3))
    a = a @ a.transpose(0, 2, 1) + 0.5 * m * np.eye(m)
    l_ref = np.linalg.cholesky(a)
    w = ops.tril_inverse_batched(T(l_ref, torch.float64, dev))
    w_ref = np.stack([np.linalg.inv(l_ref[i]) for i in range(b)])
    close(w, w_ref, dict(rtol=1e-10, atol_rel=1e-12), 'L^-1')
    assert float(torch.triu(w, 1).abs().max()) == 0.0


@pytest.mark.parametrize('m', [200, 256, 300, 512])
@pytest.mark.parametrize('dt', [torch.float64, torch.float32])
def test_potrf_large_multi_workgroup(dev, m, dt):
    """M beyond one workgroup's LDS: the right-looking 64-block factorisation spread over the GPU (potrf_big.hip)."""
    rng = np.random.default_rng(m)
    b = 5
    a = rng.standard_normal((b, m, m + 3))
    a = a @ a.transpose(0, 2, 1) + 0.5 * m * np.eye(m)
    l_ref = np.linalg.cholesky(a)
    tol = dict(rtol=1e-11, atol_rel=1e-13) if dt == torch.float64 else dict(rtol=5e-4, atol_rel=5e-6)
    l, info = ops.potrf_batched(T(a, dt, dev))
    assert int(info.abs().max()) == 0
    close(l, l_ref, tol, 'potrf (multi-workgroup)')
    bad = np.eye(m)[None].repeat(2, axis=0)
    bad[1, 150, 150] = -1.0
    _, info = ops.potrf_batched(T(bad, dt, dev))
    assert info.tolist() == [0, 151]


@pytest.mark.parametrize('m', [200, 256, 300, 384, 512, 640])
def test_potrf_large_persistent_workgroup(dev, m, monkeypatch):
    """M beyond one workgroup's LDS, the path taken when there are enough matrices to give every compute unit its own
    (B >= 128, fp64): ONE persistent workgroup per matrix (potrf_persist.hip) — forced here for a small batch through
    DPGP_POTRF_PERSISTENT=1, and once at B = 130 where the library picks it by itself."""
    monkeypatch.setenv('DPGP_POTRF_PERSISTENT', '1')
    rng = np.random.default_rng(m + 1)
    b = 3
    a = rng.standard_normal((b, m, m + 3))
    a = a @ a.transpose(0, 2, 1) + 0.5 * m * np.eye(m)
    l_ref = np.linalg.cholesky(a)
    l, info = ops.potrf_batched(T(a, torch.float64, dev))
    assert int(info.abs().max()) == 0
    close(l, l_ref, dict(rtol=1e-11, atol_rel=1e-13), 'potrf (persistent workgroup)')
    assert float(torch.triu(l, 1).abs().max()) == 0.0, 'zeros above the diagonal'
    bad = np.eye(m)[None].repeat(2, axis=0)
    bad[1, 150, 150] = -1.0
    _, info = ops.potrf_batched(T(bad, torch.float64, dev))
    assert info.tolist() == [0, 151]


def test_potrf_many_large_matrices_pick_the_persistent_path(dev, monkeypatch):
    monkeypatch.delenv('DPGP_POTRF_PERSISTENT', raising=False)
    rng = np.random.default_rng(5)
    b, m = 130, 256
    a0 = rng.standard_normal((b, m, m))
    a = a0 @ a0.transpose(0, 2, 1) + m * np.eye(m)
    l, info = ops.potrf_batched(T(a, torch.float64, dev))
    assert int(info.abs().max()) == 0
    close(l, np.linalg.cholesky(a), dict(rtol=1e-11, atol_rel=1e-13), 'potrf (B = 130)')


@pytest.mark.parametrize('algo', ['auto', 'plain'])
def test_potrf_reports_non_positive_definite(dev, algo):
    a = np.eye(40)[None].repeat(2, axis=0)
    a[1, 17, 17] = -1.0
    l, info = ops.potrf_batched(T(a, torch.float64, dev), algo=algo)
    assert info.tolist() == [0, 18]


def test_kl(dev):
    g = golden('dpgplvm_50_10_25_3_T8')
    for dt in (torch.float64, torch.float32):
        got = float(ops.kl_qx(T(g['mu'], dt, dev), T(g['s'], dt, dev)))
        np.testing.assert_allclose(got, float(g['kl']), rtol=1e-12 if dt == torch.float64 else 1e-6)


def test_operators_refuse_cpu_tensors():
    x = torch.zeros(4, 2, dtype=torch.float64)
    with pytest.raises(RuntimeError):
        ops.ard_rbf_gram(x, None, torch.ones(1, 2, dtype=torch.float64), torch.ones(1, 1, dtype=torch.float64),
                         torch.ones(1, 1, dtype=torch.float64))


@pytest.mark.parametrize('shape', [(3, 5, 7, 4), (8, 128, 128, 128), (2, 70, 33, 129), (4, 200, 1, 65), (1, 64, 64, 0),
                                   (5, 1, 96, 31)])
def test_matmul_strided_views_match_numpy(dev, shape):
    """dpgp_gemm_strided_f64 (the tf.matmul call sites of the composed models: dp_gp_lvm.py:657-658 and around) on plain,
    transposed, sliced and batch-broadcast views; fp64 against numpy's matmul of the same views (rtol 1e-13 of the row
    norms: the MFMA accumulates in a different order)."""
    b, m, n, k = shape
    rng = np.random.default_rng(7)
    a = rng.standard_normal((b, m, max(k, 1)))[:, :, :k]
    bm = rng.standard_normal((b, max(k, 1), n))[:, :k, :]
    ta, tb = T(np.ascontiguousarray(a), torch.float64, dev), T(np.ascontiguousarray(bm), torch.float64, dev)

    def check(x, y, xn, yn):
        got = ops.matmul(x, y).cpu().numpy()
        want = np.matmul(xn, yn)
        scale = np.abs(xn).sum(-1, keepdims=True).max() * max(np.abs(yn).max(), 1.0) if k else 1.0
        np.testing.assert_allclose(got, want, rtol=0, atol=1e-13 * max(scale, 1.0))

    check(ta, tb, a, bm)
    # transposed views (k-fastest / i-fastest staging on either side)
    at = T(np.ascontiguousarray(a.transpose(0, 2, 1)), torch.float64, dev)
    bt = T(np.ascontiguousarray(bm.transpose(0, 2, 1)), torch.float64, dev)
    check(at.transpose(1, 2), bt.transpose(1, 2), a, bm)
    check(at.transpose(1, 2), tb, a, bm)
    # batch broadcast of a 2-D operand on either side
    check(ta[0], tb, a[0], bm)
    check(ta, tb[0], a, bm[0])
    # sliced (non-contiguous) rows
    if m >= 4:
        check(ta[:, ::2, :], tb, a[:, ::2, :], bm)
    # out = alpha a b + beta out
    if k:
        out = T(np.ones((b, m, n)), torch.float64, dev)
        ops.matmul(ta, tb, out=out, alpha=-0.5, beta=2.0)
        np.testing.assert_allclose(out.cpu().numpy(), -0.5 * np.matmul(a, bm) + 2.0, rtol=0, atol=1e-12 * max(k, 1))


def test_matmul_refuses_host_and_fp32_tensors(dev):
    x = torch.zeros(4, 4, dtype=torch.float64)
    with pytest.raises(TypeError):
        ops.matmul(x, x)
    y = torch.zeros(4, 4, dtype=torch.float32, device=dev)
    with pytest.raises(TypeError):
        ops.matmul(y, y)
    with pytest.raises(ValueError):
        ops.matmul(torch.zeros(2, 3, dtype=torch.float64, device=dev), torch.zeros(4, 2, dtype=torch.float64, device=dev))
