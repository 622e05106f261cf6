"""
CPU tests (no GPU): the C-ABI library loads and exports every symbol include/dpgp.h declares, workspace queries (pure
host functions) behave, the operators fail loudly without a GPU, and the host-side mirror of the reference interface
(enum, Kernel wrapper, constants, PCA init, sharding helper, synthetic problems) behaves as the reference's does.
No compute kernel is called here.
"""
import os
import re

import numpy as np
import pytest
import torch

from dp_gp_lvm_amd import _lib
from dp_gp_lvm_amd.kernels.interfaces.kernel import Kernel, KernelHyperparameters
from dp_gp_lvm_amd.models.dp_gp_lvm import shard_bounds
from dp_gp_lvm_amd.utils import constants
from dp_gp_lvm_amd.utils.expressions import principal_component_analysis
from dp_gp_lvm_amd.utils.synthetic import make_problem, CONFIGS

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(REPO, 'include', 'dpgp.h')).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    return sorted(set(re.findall(r'\b(dpgp_[a-z0-9_A-Z]+)\s*\(', txt)))


def test_library_exports_every_declared_symbol():
    lib = _lib.lib()
    syms = header_symbols()
    assert len(syms) >= 30
    for s in syms:
        assert hasattr(lib, s), 'include/dpgp.h declares %s but libdpgp_hip.so does not export it' % s
    # and the ctypes table binds exactly the header
    assert sorted(_lib.SIGNATURES) == syms
    assert lib.dpgp_version() >= 100


def test_workspace_queries_are_host_only_and_monotone():
    lib = _lib.lib()
    assert lib.dpgp_elbo_workspace_bytes(0, 10, 5, 2, 1) == 0
    small = lib.dpgp_elbo_workspace_bytes(64, 2000, 128, 10, 1)
    big = lib.dpgp_elbo_workspace_bytes(512, 2000, 128, 10, 1)
    assert 0 < small < big < 2 ** 31
    assert lib.dpgp_elbo_workspace_bytes(512, 2000, 128, 10, 2) > big        # fp64 psi2 slabs are larger
    assert lib.dpgp_psi2_workspace_bytes(7, 100, 25, 10, 8) >= 7 * 32 * 32 * 8
    assert lib.dpgp_potrf_workspace_bytes(3, 20, 4) >= 3 * 32 * 32 * 4
    assert lib.dpgp_trsm_workspace_bytes(3, 20, 9, 8) > 0 and lib.dpgp_psi1T_y_workspace_bytes(4, 100, 20) > 0


def test_bad_arguments_are_reported_by_index_before_any_launch():
    lib = _lib.lib()
    assert lib.dpgp_psi2_f32(0, 1, 1, 1, None, None, None, None, None, None, None, 0, 0, None) == -1
    assert lib.dpgp_psi2_f32(1, 1, 1, 31, None, None, None, None, None, None, None, 0, 0, None) == -4     # Q > DPGP_MAX_Q
    assert lib.dpgp_psi2_f32(1, 1, 1, 1, None, None, None, None, None, None, None, 0, 0, None) == -5       # null z
    assert lib.dpgp_elbo_fhat(4, 10, 0, 2, None, 4, None, None, None, None, None, None, 1e-8, 1, 0, None, None, None,
                              None, 0, None) == -3                                                           # M <= 0
    assert lib.dpgp_elbo_fhat(4, 10, 20, 2, None, 4, None, None, None, None, None, None, 1e-8, 1, 0, None, None, None,
                              None, 0, None) == -5                                                           # M > N is legal (prediction)
    assert lib.dpgp_potrf_batched_f64(1, 0, None, None, None, 0, 0, None) == -2
    with pytest.raises(ValueError):
        _lib.check(-3, 'x')


def test_operators_fail_loudly_without_gpu():
    from dp_gp_lvm_amd import ops
    x = torch.zeros(4, 2, dtype=torch.float64)
    one = torch.ones(1, 1, dtype=torch.float64)
    with pytest.raises(RuntimeError):
        ops.psi2(x, x, x.abs() + 1, torch.ones(1, 2, dtype=torch.float64), one)
    if not torch.cuda.is_available():
        from dp_gp_lvm_amd.models.dp_gp_lvm import dp_gp_lvm
        with pytest.raises(RuntimeError):
            dp_gp_lvm(make_problem(1)['y'], num_latent_dims=4, num_inducing_points=20)


def test_missing_library_is_an_error_not_a_fallback(monkeypatch):
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'LIB_PATH', '/nonexistent/libdpgp_hip.so')
    with pytest.raises(_lib.DpgpLibraryMissing):
        _lib.lib()


def test_kernel_wrapper_contract():
    """Kernel.__init__ assertions and NotImplementedError for missing psi closures (reference kernel.py:150-165,242-246)."""
    hp = {KernelHyperparameters.ARD_WEIGHTS: torch.ones(1, 2)}
    pr = {KernelHyperparameters.ARD_WEIGHTS: lambda x: torch.zeros_like(x)}
    k = Kernel(lambda **kw: 1, lambda **kw: 2, hp, pr)
    assert k.covariance_matrix(None) == 1 and k.covariance_diag(None) == 2
    for f in (k.psi_0, k.psi_1, k.psi_2):
        with pytest.raises(NotImplementedError):
            f(None, None, None)
    with pytest.raises(AssertionError):
        Kernel(1, lambda: 0, hp, pr)
    with pytest.raises(AssertionError):
        Kernel(lambda: 0, lambda: 0, {'gamma': 1}, {'gamma': lambda x: x})
    with pytest.raises(AssertionError):
        Kernel(lambda: 0, lambda: 0, hp, {KernelHyperparameters.SIGNAL_VARIANCE: lambda x: x})
    assert [e.value for e in KernelHyperparameters] == ['gamma', 'alpha', 'beta', 'freq', 'period', 'l', 'W']


def test_constants_match_reference_values():
    assert constants.GP_DEFAULT_JITTER == 1.0e-8 and constants.GP_INIT_GAMMA == constants.GP_INIT_ALPHA == 1.0
    assert constants.GP_LVM_DEFAULT_LATENT_DIMENSIONS == 10 and constants.GP_LVM_DEFAULT_NUM_INDUCING_POINTS == 25
    assert constants.DP_DEFAULT_TRUNCATION_LEVEL == 8 and list(constants.DP_DEFAULT_ALPHA_PRIOR_PARAMS) == [1.0, 1.0]


def test_pca_init_is_deterministic_and_scaled():
    y = make_problem(1)['y']
    a = principal_component_analysis(y, 4)
    b = principal_component_analysis(y, 4)
    np.testing.assert_array_equal(a, b)                                  # the reference's ARPACK call flips signs run to run
    assert a.shape == (100, 4)
    np.testing.assert_allclose(np.mean(a.std(axis=0, ddof=1)), 1.0, rtol=1e-12)
    w = np.linalg.eigvalsh(y @ y.T)[::-1][:4]
    v = a / np.linalg.norm(a, axis=0)
    np.testing.assert_allclose(np.einsum('ij,ij->j', v, (y @ y.T) @ v), w, rtol=1e-8)
    with pytest.raises(AssertionError):
        principal_component_analysis(y, 12)


def test_shard_bounds_partition_d():
    for d, w in [(512, 8), (512, 1), (560, 8), (12, 5), (7, 7)]:
        b = [shard_bounds(d, r, w) for r in range(w)]
        assert b[0][0] == 0 and b[-1][1] == d
        assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
        sizes = [hi - lo for lo, hi in b]
        assert max(sizes) - min(sizes) <= 1 and min(sizes) > 0


def test_synthetic_problem_recipe():
    p = make_problem(2)
    n, d, m, q = CONFIGS[2]
    assert p['y'].shape == (n, d) and p['z'].shape == (m, q) and p['phi'].shape == (d, 8)
    np.testing.assert_allclose(p['y'].mean(axis=0), 0, atol=1e-12)
    np.testing.assert_allclose(p['y'].std(axis=0), 1, rtol=1e-12)
    np.testing.assert_allclose(p['phi'].sum(axis=1), 1, rtol=1e-12)
    ps = make_problem(2, d_slice=slice(16, 32))
    np.testing.assert_array_equal(ps['y'], p['y'][:, 16:32])
    np.testing.assert_array_equal(ps['gamma'], p['gamma'][16:32])
    np.testing.assert_array_equal(ps['mu'], p['mu'])


def test_stage_b_workspace_is_sized_for_the_form_that_runs():
    """dpgp_elbo_grad_psi_workspace_bytes_ex (host-only): the images and results of the pair-tile form (~2 GB at BASELINE config 3)
    are part of the workspace only for the precisions that run it; the patch form asked for explicitly (the training configuration
    behind an fp64 forward pass) stays under 0.5 GB there; fp64 needs the plain kernel's partials only; the query without a precision
    is the largest of them."""
    from dp_gp_lvm_amd import _lib
    l = _lib.lib()
    n, d, m, q = 2000, 512, 128, 10
    size = {p: int(l.dpgp_elbo_grad_psi_workspace_bytes_ex(d, n, m, q, _lib.PREC[p])) for p in ('mixed', 'mixed_patch', 'mixed_fast', 'f64')}
    assert size['mixed_patch'] < 0.5e9 < size['mixed'] == size['mixed_fast']
    assert size['f64'] < size['mixed_patch']
    assert int(l.dpgp_elbo_grad_psi_workspace_bytes(d, n, m, q)) == max(size.values())
    # BASELINE config 4 (M = 512, Q = 20: the pair-tile form with two feature blocks): no patch-form partials beside it
    assert int(l.dpgp_elbo_grad_psi_workspace_bytes_ex(256, 10000, 512, 20, _lib.PREC['mixed'])) < 30e9
