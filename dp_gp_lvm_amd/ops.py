"""
Torch-tensor front-end of the C ABI (include/dpgp.h).  PyTorch is plumbing here: device memory, the current HIP stream
and (in models/) torch.distributed.  Every function takes CUDA (ROCm) tensors, enqueues HIP kernels on the current
stream and returns device tensors; nothing here computes on the host and there is no fallback.
"""
import torch

from . import _lib

_SUFFIX = {torch.float32: 'f32', torch.float64: 'f64'}


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _prep(t, dtype, name):
    if not isinstance(t, torch.Tensor):
        raise TypeError('%s must be a torch.Tensor' % name)
    if not t.is_cuda:
        raise RuntimeError('%s must live on the GPU: dp_gp_lvm_amd runs its operators in HIP only' % name)
    if t.dtype != dtype:
        t = t.to(dtype)
    return t.contiguous()


def _dtype_of(*ts):
    dt = ts[0].dtype
    if dt not in _SUFFIX:
        raise TypeError('expected float32 or float64 tensors, got %s' % dt)
    return dt


def _ws(nbytes, device):
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


def _hyp(gamma, alpha, beta, dt):
    gamma = _prep(gamma, dt, 'gamma')
    assert gamma.dim() == 2, 'gamma must be [B x Q]'
    b = gamma.shape[0]
    alpha = _prep(alpha, dt, 'alpha').reshape(-1)
    assert alpha.numel() == b, 'alpha must be [B x 1]'
    if beta is not None:
        beta = _prep(beta, dt, 'beta').reshape(-1)
        assert beta.numel() == b, 'beta must be [B x 1]'
    return gamma, alpha, beta, b


def ard_rbf_gram(x0, x1, gamma, alpha, beta, include_noise=False, include_jitter=False, jitter=1e-8):
    """Kernel.covariance_matrix -> [B,N0,N1]  (reference: src/kernels/rbf_kernel.py:58-93)."""
    dt = _dtype_of(x0)
    x0 = _prep(x0, dt, 'input_0')
    gamma, alpha, beta, b = _hyp(gamma, alpha, beta, dt)
    q = gamma.shape[1]
    assert x0.dim() == 2 and x0.shape[1] == q, 'input_0 must be [N0 x Q]'
    n0 = x0.shape[0]
    n1 = n0
    x1p = None
    if x1 is not None:
        x1 = _prep(x1, dt, 'input_1')
        assert x1.dim() == 2 and x1.shape[1] == q, 'input_1 must be [N1 x Q]'
        n1, x1p = x1.shape[0], x1.data_ptr()
    out = torch.empty((b, n0, n1), dtype=dt, device=x0.device)
    flags = (_lib.FLAG_NOISE if include_noise else 0) | (_lib.FLAG_JITTER if include_jitter else 0)
    name = 'dpgp_ard_rbf_gram_' + _SUFFIX[dt]
    _lib.check(getattr(_lib.lib(), name)(b, n0, n1, q, x0.data_ptr(), x1p, gamma.data_ptr(), alpha.data_ptr(),
                                         beta.data_ptr(), flags, float(jitter), out.data_ptr(), _stream()), name)
    return out


def ard_rbf_diag(n, alpha, beta, include_noise=False, include_jitter=False, jitter=1e-8):
    """Kernel.covariance_diag -> [B,N]  (rbf_kernel.py:96-116)."""
    dt = _dtype_of(alpha)
    alpha = _prep(alpha, dt, 'alpha').reshape(-1)
    beta = _prep(beta, dt, 'beta').reshape(-1)
    b = alpha.numel()
    out = torch.empty((b, int(n)), dtype=dt, device=alpha.device)
    flags = (_lib.FLAG_NOISE if include_noise else 0) | (_lib.FLAG_JITTER if include_jitter else 0)
    name = 'dpgp_ard_rbf_diag_' + _SUFFIX[dt]
    _lib.check(getattr(_lib.lib(), name)(b, int(n), alpha.data_ptr(), beta.data_ptr(), flags, float(jitter),
                                         out.data_ptr(), _stream()), name)
    return out


def psi0(n, alpha):
    """Kernel.psi_0 -> [B,1]  (rbf_kernel.py:119-132)."""
    dt = _dtype_of(alpha)
    alpha = _prep(alpha, dt, 'alpha').reshape(-1)
    out = torch.empty((alpha.numel(), 1), dtype=dt, device=alpha.device)
    name = 'dpgp_psi0_' + _SUFFIX[dt]
    _lib.check(getattr(_lib.lib(), name)(alpha.numel(), int(n), alpha.data_ptr(), out.data_ptr(), _stream()), name)
    return out


def _zms(z, mu, s, gamma, alpha):
    dt = _dtype_of(mu)
    z, mu, s = _prep(z, dt, 'inducing_input'), _prep(mu, dt, 'latent_input_mean'), _prep(s, dt, 'latent variance')
    gamma, alpha, _, b = _hyp(gamma, alpha, None, dt)
    q = gamma.shape[1]
    assert z.dim() == 2 and z.shape[1] == q, 'inducing_input must be [M x Q]'
    assert mu.dim() == 2 and mu.shape[1] == q, 'latent_input_mean must be [N x Q]'
    assert s.shape == mu.shape, 'latent variances must be [N x Q]'
    return dt, z, mu, s, gamma, alpha, b, mu.shape[0], z.shape[0], q


def psi1(z, mu, s, gamma, alpha):
    """Kernel.psi_1 -> [B,N,M]  (rbf_kernel.py:135-161)."""
    dt, z, mu, s, gamma, alpha, b, n, m, q = _zms(z, mu, s, gamma, alpha)
    out = torch.empty((b, n, m), dtype=dt, device=mu.device)
    name = 'dpgp_psi1_' + _SUFFIX[dt]
    _lib.check(getattr(_lib.lib(), name)(b, n, m, q, z.data_ptr(), mu.data_ptr(), s.data_ptr(), gamma.data_ptr(),
                                         alpha.data_ptr(), out.data_ptr(), _stream()), name)
    return out


def psi1T_y(z, mu, s, gamma, alpha, y):
    """Psi1_b^T y_b -> [B,M] without materialising Psi1 (dp_gp_lvm.py:132-145)."""
    dt, z, mu, s, gamma, alpha, b, n, m, q = _zms(z, mu, s, gamma, alpha)
    y = _prep(y, dt, 'y')
    assert y.shape == (n, b), 'y must be [N x B]'
    out = torch.empty((b, m), dtype=dt, device=mu.device)
    l = _lib.lib()
    wsb = l.dpgp_psi1T_y_workspace_bytes(b, n, m)
    ws = _ws(wsb, mu.device)
    name = 'dpgp_psi1T_y_' + _SUFFIX[dt]
    _lib.check(getattr(l, name)(b, n, m, q, z.data_ptr(), mu.data_ptr(), s.data_ptr(), gamma.data_ptr(),
                                alpha.data_ptr(), y.data_ptr(), y.stride(0), out.data_ptr(), ws.data_ptr(), wsb,
                                _stream()), name)
    return out


def psi2(z, mu, s, gamma, alpha, algo='auto'):
    """Kernel.psi_2 -> [B,M,M]  (rbf_kernel.py:164-199), streamed over n on the matrix cores."""
    dt, z, mu, s, gamma, alpha, b, n, m, q = _zms(z, mu, s, gamma, alpha)
    out = torch.empty((b, m, m), dtype=dt, device=mu.device)
    l = _lib.lib()
    wsb = l.dpgp_psi2_workspace_bytes(b, n, m, q, out.element_size())
    ws = _ws(wsb, mu.device)
    name = 'dpgp_psi2_' + _SUFFIX[dt]
    _lib.check(getattr(l, name)(b, n, m, q, z.data_ptr(), mu.data_ptr(), s.data_ptr(), gamma.data_ptr(),
                                alpha.data_ptr(), out.data_ptr(), ws.data_ptr(), wsb, _lib.ALGO[algo], _stream()), name)
    return out


def potrf_batched(a, algo='auto'):
    """tf.cholesky on [B,M,M] -> (L [B,M,M] lower, info [B] int32)  (dp_gp_lvm.py:116,127)."""
    dt = _dtype_of(a)
    assert a.dim() == 3 and a.shape[1] == a.shape[2], 'a must be [B x M x M]'
    l_ = _prep(a, dt, 'a').clone()
    b, m = l_.shape[0], l_.shape[1]
    info = torch.empty(b, dtype=torch.int32, device=a.device)
    l = _lib.lib()
    wsb = l.dpgp_potrf_workspace_bytes(b, m, l_.element_size())
    ws = _ws(wsb, a.device)
    name = 'dpgp_potrf_batched_' + _SUFFIX[dt]
    _lib.check(getattr(l, name)(b, m, l_.data_ptr(), info.data_ptr(), ws.data_ptr(), wsb, _lib.ALGO[algo], _stream()),
               name)
    return l_, info


def trsm_batched(l_, rhs, algo='auto'):
    """tf.matrix_triangular_solve(l, rhs, lower=True) on [B,M,M], [B,M,K] -> [B,M,K]  (dp_gp_lvm.py:118-121)."""
    dt = _dtype_of(l_)
    l_ = _prep(l_, dt, 'l')
    x = _prep(rhs, dt, 'rhs').clone()
    assert l_.dim() == 3 and x.dim() == 3 and l_.shape[0] == x.shape[0] and l_.shape[1] == l_.shape[2] == x.shape[1]
    b, m, k = x.shape
    l = _lib.lib()
    wsb = l.dpgp_trsm_workspace_bytes(b, m, k, x.element_size())
    ws = _ws(wsb, x.device)
    name = 'dpgp_trsm_batched_' + _SUFFIX[dt]
    _lib.check(getattr(l, name)(b, m, k, l_.data_ptr(), x.data_ptr(), ws.data_ptr(), wsb, _lib.ALGO[algo], _stream()),
               name)
    return x


def tril_inverse_batched(l_):
    """tf.matrix_triangular_solve(l, eye, lower=True) on [B,M,M] fp64 -> L^-1 [B,M,M] (lower, zeros above the diagonal).  M a multiple
    of 128: one persistent-workgroup launch (dpgp_trtri_lower_batched_f64); otherwise dpgp_trsm_batched on the identity."""
    f64 = torch.float64
    l_ = _prep(l_, f64, 'l')
    b, m = l_.shape[0], l_.shape[1]
    assert l_.dim() == 3 and l_.shape[2] == m
    if m % 128 != 0:
        return trsm_batched(l_, torch.eye(m, dtype=f64, device=l_.device).expand(b, m, m).contiguous())
    out = torch.empty_like(l_)
    ws = _ws(8 * b, l_.device)
    _lib.check(_lib.lib().dpgp_trtri_lower_batched_f64(b, m, l_.data_ptr(), out.data_ptr(), ws.data_ptr(), 8 * b, _stream()),
               'dpgp_trtri_lower_batched_f64')
    return out


def matmul(a, b, out=None, alpha=1.0, beta=0.0):
    """tf.matmul on fp64 device tensors of 2 or 3 dimensions (batch broadcast as in TensorFlow / torch.matmul), through the
    library's strided batched MFMA kernel (dpgp_gemm_strided_f64): transposed or sliced VIEWS are passed by their strides, no
    copies and no rocBLAS.  The reference's call sites: dp_gp_lvm.py:657-658 and the composed chains around it.
    out (optional, [batch, m, n] or [m, n]): out = alpha a b + beta out."""
    f64 = torch.float64
    for t, name in ((a, 'a'), (b, 'b')):
        if not isinstance(t, torch.Tensor) or not t.is_cuda or t.dtype != f64:
            raise TypeError('%s must be a float64 tensor on the GPU (dp_gp_lvm_amd has no host path)' % name)
        if t.dim() not in (2, 3):
            raise ValueError('%s must have 2 or 3 dimensions' % name)
    m, k, n = a.shape[-2], a.shape[-1], b.shape[-1]
    if b.shape[-2] != k:
        raise ValueError('inner dimensions differ: %s x %s' % (tuple(a.shape), tuple(b.shape)))
    ba = a.shape[0] if a.dim() == 3 else 1
    bb = b.shape[0] if b.dim() == 3 else 1
    if ba != bb and ba != 1 and bb != 1:
        raise ValueError('batch dimensions differ: %s x %s' % (tuple(a.shape), tuple(b.shape)))
    batch = max(ba, bb)
    three = a.dim() == 3 or b.dim() == 3
    if out is None:
        if beta != 0.0:
            raise ValueError('beta needs out')
        out = torch.empty((batch, m, n) if three else (m, n), dtype=f64, device=a.device)
    else:
        assert out.dtype == f64 and out.is_cuda and tuple(out.shape) == ((batch, m, n) if three else (m, n))
    if m == 0 or n == 0:
        return out
    a_sb = a.stride(0) if (a.dim() == 3 and ba > 1) else 0
    b_sb = b.stride(0) if (b.dim() == 3 and bb > 1) else 0
    c_sb = out.stride(0) if three else 0
    _lib.check(_lib.lib().dpgp_gemm_strided_f64(batch, m, n, k, float(alpha), a.data_ptr(), a_sb, a.stride(-2), a.stride(-1),
                                                b.data_ptr(), b_sb, b.stride(-2), b.stride(-1), float(beta), out.data_ptr(),
                                                c_sb, out.stride(-2), out.stride(-1), _stream()), 'dpgp_gemm_strided_f64')
    return out


def kl_qx(mu, s):
    """calculate_kl_divergence_standard_prior -> 0-d fp64 tensor  (gp_expressions.py:10-24)."""
    dt = _dtype_of(mu)
    mu, s = _prep(mu, dt, 'x_mean'), _prep(s, dt, 'x_var')
    assert mu.shape == s.shape and mu.dim() == 2
    out = torch.empty(1, dtype=torch.float64, device=mu.device)
    name = 'dpgp_kl_qx_' + _SUFFIX[dt]
    _lib.check(getattr(_lib.lib(), name)(mu.shape[0], mu.shape[1], mu.data_ptr(), s.data_ptr(), out.data_ptr(),
                                         _stream()), name)
    return out[0]


class ElboWorkspace:
    """Caller-owned scratch + outputs of the fused ELBO for one (D,N,M,Q,prec): allocate once, evaluate many times."""

    def __init__(self, d, n, m, q, prec='mixed', device='cuda'):
        self.shape, self.prec = (d, n, m, q), prec
        l = _lib.lib()
        self.nbytes = l.dpgp_elbo_workspace_bytes(d, n, m, q, _lib.PREC[prec])
        self.ws = _ws(self.nbytes, device)
        # (before the first evaluation: NaN terms and zero flags, not whatever the allocator hands out)
        self.terms = torch.full((d, 5), float('nan'), dtype=torch.float64, device=device)
        self.sums = torch.full((2,), float('nan'), dtype=torch.float64, device=device)
        self.info = torch.zeros(d, dtype=torch.int32, device=device)
        self.exec = _lib.ExecResources()
        # second stream + fork / join events: Psi1^T y beside the psi2 launch (include/dpgp.h, dpgp_exec_t.stream_aux).
        # Opt-in (DPGP_PARALLEL_BRANCH=1): measured on MI355X it is SLOWER than one stream — config 2: 0.263 vs 0.253 ms per
        # evaluation, config 3: 1.371 vs 1.298, replayed from a HIP graph 0.345 — the two event hand-offs cost more than the
        # 12 us (D = 64) the branch could hide, and the psi2 launch fills the chip on its own.
        import os
        if os.environ.get('DPGP_PARALLEL_BRANCH', '0') == '1':
            self.exec.stream_aux, self.exec.ev_fork, self.exec.ev_join = l.dpgp_stream_create(), l.dpgp_event_create(), l.dpgp_event_create()
        import ctypes
        lay = (ctypes.c_size_t * 10)()
        _lib.check(l.dpgp_elbo_workspace_layout(d, n, m, q, _lib.PREC[prec], ctypes.cast(lay, ctypes.c_void_p)),
                   'dpgp_elbo_workspace_layout')
        self.layout = tuple(int(v) for v in lay)
        # guard[d]: bound on what the rounding of an fp32 Psi2 can do to output dim d's terms (include/dpgp.h,
        # DPGP_INFO_ILL_CONDITIONED); a view into the workspace, valid after an evaluation
        self.guard = self.ws[self.layout[8]:self.layout[8] + 8 * d].view(torch.float64)


    def __del__(self):
        try:
            l = _lib.lib()
            if self.exec.stream_aux:
                l.dpgp_stream_destroy(self.exec.stream_aux)
                l.dpgp_event_destroy(self.exec.ev_fork)
                l.dpgp_event_destroy(self.exec.ev_join)
                self.exec.stream_aux = None
        except Exception:                                        # (interpreter shutdown)
            pass


def elbo_fhat(y, z, mu, s, gamma, alpha, beta, jitter=1e-8, prec='mixed', algo='auto', workspace=None, events=None,
              model_tail=None):
    """
    The fused per-output ELBO reduction of dp_gp_lvm.py:108-148 for the D output dims in ``y`` [N,D] (a column slice of
    the full data when D is sharded over GPUs).  All inputs fp64 device tensors.
    Returns (terms [D,5], sums [2] = (f_hat, KL), info [D]) — device tensors, no host synchronisation.
    """
    f64 = torch.float64
    z, mu, s = _prep(z, f64, 'z'), _prep(mu, f64, 'mu'), _prep(s, f64, 's')
    gamma, alpha, beta, d = _hyp(gamma, alpha, beta, f64)
    if not y.is_cuda:
        raise RuntimeError('y must live on the GPU')
    if y.dtype != f64 or y.stride(1) != 1:
        y = y.to(f64).contiguous()
    n, m, q = mu.shape[0], z.shape[0], z.shape[1]
    assert y.shape == (n, d), 'y must be [N x D]'
    assert mu.shape == (n, q) and s.shape == (n, q) and gamma.shape == (d, q)
    w = workspace if workspace is not None else ElboWorkspace(d, n, m, q, prec, y.device)
    assert w.shape == (d, n, m, q) and w.prec == prec, 'workspace was sized for another problem'
    ev0, ev1 = events if events is not None else (None, None)     # optional hipEvent handles around the psi2 kernel
    w.exec.ev_psi2_begin, w.exec.ev_psi2_end = ev0, ev1
    # model_tail = (scal, pack, out) device tensors (out may be None): fold dpgp_model_pack / _finalize into the last launch
    w.exec.model_scal, w.exec.model_pack, w.exec.model_out = (
        (None, None, None) if model_tail is None else
        tuple(None if t_ is None else t_.data_ptr() for t_ in model_tail))
    import ctypes
    _lib.check(_lib.lib().dpgp_elbo_fhat_ex(d, n, m, q, y.data_ptr(), y.stride(0), z.data_ptr(), mu.data_ptr(),
                                            s.data_ptr(), gamma.data_ptr(), alpha.data_ptr(), beta.data_ptr(),
                                            float(jitter), _lib.PREC[prec], _lib.ALGO[algo], w.terms.data_ptr(),
                                            w.sums.data_ptr(), w.info.data_ptr(), w.ws.data_ptr(), w.nbytes,
                                            _stream(), ctypes.cast(ctypes.pointer(w.exec), ctypes.c_void_p)),
               'dpgp_elbo_fhat_ex')
    return w.terms, w.sums, w.info


class ElboTWorkspace:
    """Device buffers of ``elbo_fhat_t`` for one problem shape (allocated once, reused by every evaluation)."""

    def __init__(self, t, d, n, m, q, prec, device):
        nbytes = int(_lib.lib().dpgp_elbo_fhat_t_workspace_bytes(t, d, n, m, q, _lib.PREC[prec]))
        if nbytes == 0:
            raise ValueError('elbo_fhat_t: unsupported shape / precision (T=%d D=%d N=%d M=%d Q=%d %s)' % (t, d, n, m, q, prec))
        self.shape, self.prec, self.nbytes = (t, d, n, m, q), prec, nbytes
        self.ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
        self.per_t = torch.empty(t, dtype=torch.float64, device=device)
        self.quad = torch.empty((t, d), dtype=torch.float64, device=device)
        self.sums = torch.empty(2, dtype=torch.float64, device=device)
        self.info = torch.empty(t, dtype=torch.int32, device=device)
        # second stream + fork / join events: Psi1 and Psi1^T Y beside the fused reduction on the atoms (include/dpgp.h);
        # DPGP_PARALLEL_BRANCH_T=0 keeps one stream
        import os
        l = _lib.lib()
        self.aux = (l.dpgp_stream_create(), l.dpgp_event_create(), l.dpgp_event_create()) \
            if os.environ.get('DPGP_PARALLEL_BRANCH_T', '1') != '0' else (None, None, None)

    def __del__(self):
        try:
            if self.aux[0]:
                l = _lib.lib()
                l.dpgp_stream_destroy(self.aux[0])
                l.dpgp_event_destroy(self.aux[1])
                l.dpgp_event_destroy(self.aux[2])
                self.aux = (None, None, None)
        except Exception:
            pass


def elbo_fhat_t_supported(m):
    """dpgp_elbo_fhat_t keeps the factors L_B,t in LDS (first version): M <= 128."""
    return 16 * ((int(m) + 15) // 16) <= 128


def elbo_fhat_t(y, yy, z, mu, s, gamma_atoms, alpha_atoms, beta_atoms, phit, jitter=1e-8, prec='mixed', workspace=None,
                model_tail=None):
    """
    f_hat of the over-T model (reference dp_gp_lvm.py:608-676) for the D output dims in ``y`` [N,D]: T atoms, each solved against
    all D columns, weighted by ``phit`` [T,D] (any strides: phi[D,T].t() is read in place).  ``yy`` [D] = column sums of y^2.
    All inputs fp64 device tensors.  model_tail = (scal, pack, out) as for ``elbo_fhat``.
    Returns (per_t [T], quad [T,D], sums [2] = (f_hat, KL), info [T]) — device tensors, no host synchronisation (nine launches).
    """
    f64 = torch.float64
    z, mu, s = _prep(z, f64, 'z'), _prep(mu, f64, 'mu'), _prep(s, f64, 's')
    gamma, alpha, beta, t = _hyp(gamma_atoms, alpha_atoms, beta_atoms, f64)
    if not y.is_cuda:
        raise RuntimeError('y must live on the GPU')
    if y.dtype != f64 or y.stride(1) != 1:
        y = y.to(f64).contiguous()
    n, d = y.shape
    m, q = z.shape
    yy = _prep(yy, f64, 'yy').reshape(-1)
    if not phit.is_cuda or phit.dtype != f64:
        raise RuntimeError('phit must be an fp64 GPU tensor')
    assert mu.shape == (n, q) and s.shape == (n, q) and gamma.shape == (t, q)
    assert yy.numel() == d and phit.shape == (t, d), 'yy must be [D], phit [T x D]'
    w = workspace if workspace is not None else ElboTWorkspace(t, d, n, m, q, prec, y.device)
    assert w.shape == (t, d, n, m, q) and w.prec == prec, 'workspace was sized for another problem'
    tail = (None, None, None) if model_tail is None else tuple(None if t_ is None else t_.data_ptr() for t_ in model_tail)
    _lib.check(_lib.lib().dpgp_elbo_fhat_t(t, d, n, m, q, y.data_ptr(), y.stride(0), yy.data_ptr(), z.data_ptr(), mu.data_ptr(),
                                           s.data_ptr(), gamma.data_ptr(), alpha.data_ptr(), beta.data_ptr(), phit.data_ptr(),
                                           phit.stride(0), phit.stride(1), float(jitter), _lib.PREC[prec], w.per_t.data_ptr(),
                                           w.quad.data_ptr(), w.sums.data_ptr(), w.info.data_ptr(), w.ws.data_ptr(), w.nbytes,
                                           _stream(), *tail, *w.aux),
               'dpgp_elbo_fhat_t')
    return w.per_t, w.quad, w.sums, w.info


def elbo_step_supported(m, q):
    """dpgp_elbo_step: B_d in LDS (M <= 128) and the pair-tile form of stage B (Q <= 20)."""
    return 16 * ((int(m) + 15) // 16) <= 128 and int(q) <= 20


class ElboStepBuffers:
    """Outputs and stage-B workspace of ``elbo_step`` for one problem shape (allocated once: the workspace holds the operand images
    and the results of the two passes, ~1.6 GB at N=2000, D=512, M=128, Q=10)."""

    def __init__(self, d, n, m, q, device):
        f64 = torch.float64
        mp = 16 * ((m + 15) // 16)
        self.shape = (d, n, m, q)
        if mp <= 128:                                            # (stage A outputs of the one-call step; M > 128: the caller's)
            self.gp = torch.empty((d, mp, mp), dtype=f64, device=device)
            self.wk = torch.empty((d, mp, mp), dtype=f64, device=device)
            self.gv = torch.empty((d, mp), dtype=f64, device=device)
            self.dab = torch.empty((d, 2), dtype=f64, device=device)
            self.info = torch.empty(d, dtype=torch.int32, device=device)
        self.nbytes = int(_lib.lib().dpgp_elbo_grad_psi_workspace_bytes_ex(d, n, m, q, _lib.PREC['mixed']))
        self.ws = _ws(self.nbytes, device)
        self.dmu, self.ds = torch.empty((n, q), dtype=f64, device=device), torch.empty((n, q), dtype=f64, device=device)
        self.dz, self.dg = torch.empty((m, q), dtype=f64, device=device), torch.empty((d, q), dtype=f64, device=device)
        # second stream + fork / join events of the one-call step: Psi1^T y and the K_uu branch (in a step the latter is a launch of its
        # own) run beside the image build and the head of pass 1.  DPGP_STEP_AUX=0 keeps one stream.
        import os
        self.aux = None
        if mp <= 128 and os.environ.get('DPGP_STEP_AUX', '1') != '0':
            l = _lib.lib()
            self.aux = (l.dpgp_stream_create(), l.dpgp_event_create(), l.dpgp_event_create())

    def __del__(self):
        try:
            if self.aux:
                l = _lib.lib()
                l.dpgp_stream_destroy(self.aux[0])
                l.dpgp_event_destroy(self.aux[1])
                l.dpgp_event_destroy(self.aux[2])
                self.aux = None
        except Exception:                                        # (interpreter shutdown)
            pass


def elbo_step(y, z, mu, s, gamma, alpha, beta, workspace, buffers, jitter=1e-8, model_tail=None, stage_b='mixed'):
    """One training step's worth of the fused reduction in mixed precision (dpgp_elbo_step): the f_hat terms of ``elbo_fhat``
    and the gradients of ``elbo_grad_chain`` + ``elbo_grad_psi`` from ONE call, the Psi2 exponentials evaluated twice instead of
    three times (dp_gp_lvm.py:108-145 and its tf.gradients, test/synthetic_data_hard_test.py:143-155).
    workspace: ElboWorkspace(..., 'mixed'); buffers: ElboStepBuffers.  All inputs fp64 device tensors (as ``elbo_fhat``).
    stage_b: 'mixed', or 'mixed_fast' (DPGP_PREC_MIXED_FAST, include/dpgp.h: 11-bit exponentials in the second products of stage B).
    Returns (terms, sums, info), (d_mu, d_s, d_z, d_gamma, d_alpha_beta, info_grad) — tensors of the two buffer objects."""
    import ctypes
    w, b = workspace, buffers
    d, n, m, q = w.shape
    assert b.shape == w.shape and w.prec == 'mixed', 'buffers were sized for another problem'
    assert y.is_cuda and y.dtype == torch.float64 and y.stride(1) == 1 and y.shape == (n, d)
    w.exec.ev_psi2_begin, w.exec.ev_psi2_end = None, None
    w.exec.model_scal, w.exec.model_pack, w.exec.model_out = (
        (None, None, None) if model_tail is None else tuple(None if t_ is None else t_.data_ptr() for t_ in model_tail))
    own = (w.exec.stream_aux, w.exec.ev_fork, w.exec.ev_join)
    if b.aux and not own[0]:
        w.exec.stream_aux, w.exec.ev_fork, w.exec.ev_join = b.aux
    try:
        _lib.check(_lib.lib().dpgp_elbo_step(
            d, n, m, q, y.data_ptr(), y.stride(0), z.data_ptr(), mu.data_ptr(), s.data_ptr(), gamma.data_ptr(), alpha.data_ptr(),
            beta.data_ptr(), float(jitter), _lib.PREC[stage_b], w.terms.data_ptr(), w.sums.data_ptr(), w.info.data_ptr(), w.ws.data_ptr(),
            w.nbytes, b.gp.data_ptr(), b.wk.data_ptr(), b.gv.data_ptr(), b.dab.data_ptr(), b.info.data_ptr(), b.ws.data_ptr(), b.nbytes,
            b.dmu.data_ptr(), b.ds.data_ptr(), b.dz.data_ptr(), b.dg.data_ptr(), _stream(),
            ctypes.cast(ctypes.pointer(w.exec), ctypes.c_void_p)), 'dpgp_elbo_step')
    finally:
        w.exec.stream_aux, w.exec.ev_fork, w.exec.ev_join = own
    return (w.terms, w.sums, w.info), (b.dmu, b.ds, b.dz, b.dg, b.dab, b.info)


def elbo_fhat_step(y, z, mu, s, gamma, alpha, beta, workspace, buffers, jitter=1e-8, model_tail=None):
    """The forward half of ``elbo_step`` as a call of its own (dpgp_elbo_fhat_step; any M): as ``elbo_fhat`` in mixed precision, Psi2
    out of the first pass of stage B, whose results stay in ``buffers.ws`` for ``elbo_grad_psi_step``.  The forward workspace then
    holds ONE Psi2 slab (``elbo_grad_chain(..., psi2_slabs=1)``)."""
    import ctypes
    w, b = workspace, buffers
    d, n, m, q = w.shape
    assert b.shape == w.shape and w.prec == 'mixed'
    assert y.is_cuda and y.dtype == torch.float64 and y.stride(1) == 1 and y.shape == (n, d)
    w.exec.ev_psi2_begin, w.exec.ev_psi2_end = None, None
    w.exec.model_scal, w.exec.model_pack, w.exec.model_out = (
        (None, None, None) if model_tail is None else tuple(None if t_ is None else t_.data_ptr() for t_ in model_tail))
    _lib.check(_lib.lib().dpgp_elbo_fhat_step(
        d, n, m, q, y.data_ptr(), y.stride(0), z.data_ptr(), mu.data_ptr(), s.data_ptr(), gamma.data_ptr(), alpha.data_ptr(),
        beta.data_ptr(), float(jitter), w.terms.data_ptr(), w.sums.data_ptr(), w.info.data_ptr(), w.ws.data_ptr(), w.nbytes,
        b.ws.data_ptr(), b.nbytes, _stream(), ctypes.cast(ctypes.pointer(w.exec), ctypes.c_void_p)), 'dpgp_elbo_fhat_step')
    return w.terms, w.sums, w.info


def elbo_grad_psi_step(y, z, mu, s, gamma, alpha, g_psi2, w_kuu, g_v, workspace, buffers, stage_b='mixed'):
    """The rest of stage B after ``elbo_fhat_step`` on the same workspace / buffers (dpgp_elbo_grad_psi_step)."""
    w, b = workspace, buffers
    d, n, m, q = w.shape
    _lib.check(_lib.lib().dpgp_elbo_grad_psi_step(
        d, n, m, q, y.data_ptr(), y.stride(0), z.data_ptr(), mu.data_ptr(), s.data_ptr(), gamma.data_ptr(), alpha.data_ptr(),
        g_psi2.data_ptr(), w_kuu.data_ptr(), g_v.data_ptr(), _lib.PREC[stage_b], w.ws.data_ptr(), w.nbytes, b.ws.data_ptr(), b.nbytes,
        b.dmu.data_ptr(), b.ds.data_ptr(), b.dz.data_ptr(), b.dg.data_ptr(), _stream()), 'dpgp_elbo_grad_psi_step')
    return b.dmu, b.ds, b.dz, b.dg


def elbo_grad_chain(alpha, beta, workspace, jitter=1e-8, z=None, gamma=None, psi2_slabs=None):
    """Backward pass, stage A: adjoints of the per-output dense algebra from the workspace of a finished ``elbo_fhat`` call
    (dp_gp_lvm.py:108-145 differentiated; prec mixed / f64).  M <= 128: one HIP kernel per output dim with B in LDS
    (dpgp_elbo_grad_chain).  M > 128 (needs z and gamma, mixed only downstream): composed here from the library's batched
    Cholesky / triangular solves and plain fp64 GEMMs (``_elbo_grad_chain_large``).
    psi2_slabs: the number of Psi2 partial slabs the forward evaluation left (1 after ``elbo_fhat_step``; default: the layout's).
    Returns (g_psi2 [D,Mp,Mp], w_kuu [D,Mp,Mp], g_v [D,Mp], d_alpha_beta [D,2], info [D]); see include/dpgp.h."""
    f64 = torch.float64
    d, n, m, q = workspace.shape
    alpha = _prep(alpha, f64, 'alpha').reshape(-1)
    beta = _prep(beta, f64, 'beta').reshape(-1)
    assert alpha.numel() == d and beta.numel() == d
    mp = 16 * ((m + 15) // 16)
    if mp > 128 and z is not None and gamma is not None:
        return _elbo_grad_chain_large(alpha, beta, workspace, jitter, z, gamma, psi2_slabs)
    dev = workspace.ws.device
    gp = torch.empty((d, mp, mp), dtype=f64, device=dev)
    wk = torch.empty((d, mp, mp), dtype=f64, device=dev)
    gv = torch.empty((d, mp), dtype=f64, device=dev)
    dab = torch.empty((d, 2), dtype=f64, device=dev)
    info = torch.empty(d, dtype=torch.int32, device=dev)
    _lib.check(_lib.lib().dpgp_elbo_grad_chain(d, n, m, q, alpha.data_ptr(), beta.data_ptr(), float(jitter),
                                               _lib.PREC[workspace.prec], workspace.ws.data_ptr(), workspace.nbytes,
                                               gp.data_ptr(), wk.data_ptr(), gv.data_ptr(), dab.data_ptr(), info.data_ptr(),
                                               _stream()), 'dpgp_elbo_grad_chain')
    return gp, wk, gv, dab, info


def _elbo_grad_chain_large(alpha, beta, workspace, jitter, z, gamma, psi2_slabs=None):
    """Stage A for M > 128.  M a multiple of 128: ONE library call, dpgp_elbo_grad_chain_big (csrc/chain_grad_big.hip: persistent Cholesky
    and solve, five strided MFMA products, three streaming kernels); other M (or DPGP_STAGE_A_COMPOSED=1, the cross-check): composed here
    from the batched operators (``_elbo_grad_chain_large_composed``)."""
    import os
    d, n, m, q = workspace.shape
    if m % 128 != 0 or workspace.prec not in ('mixed', 'f64') or os.environ.get('DPGP_STAGE_A_COMPOSED', '0') == '1':
        return _elbo_grad_chain_large_composed(alpha, beta, workspace, jitter, z, gamma, psi2_slabs)
    f64 = torch.float64
    dev = workspace.ws.device
    l = _lib.lib()
    z, gamma = _prep(z, f64, 'z'), _prep(gamma, f64, 'gamma')
    wsb = int(l.dpgp_elbo_grad_chain_big_workspace_bytes(d, m))
    ws = _ws(wsb, dev)
    gp, wk = torch.empty((d, m, m), dtype=f64, device=dev), torch.empty((d, m, m), dtype=f64, device=dev)
    gv, dab = torch.empty((d, m), dtype=f64, device=dev), torch.empty((d, 2), dtype=f64, device=dev)
    info = torch.empty(d, dtype=torch.int32, device=dev)
    _lib.check(l.dpgp_elbo_grad_chain_big(d, n, m, q, z.data_ptr(), gamma.data_ptr(), alpha.data_ptr(), beta.data_ptr(), float(jitter),
                                          _lib.PREC[workspace.prec], workspace.ws.data_ptr(), workspace.nbytes,
                                          0 if psi2_slabs is None else int(psi2_slabs), ws.data_ptr(), wsb, gp.data_ptr(), wk.data_ptr(),
                                          gv.data_ptr(), dab.data_ptr(), info.data_ptr(), _stream()), 'dpgp_elbo_grad_chain_big')
    return gp, wk, gv, dab, info


def _elbo_grad_chain_large_composed(alpha, beta, workspace, jitter, z, gamma, psi2_slabs=None):
    """Stage A for M > 128 (first version): the same adjoints as chain_grad_kernel (grad.hip), with B^-1 and K^-1 formed
    explicitly from the library's Cholesky factors — L^-1 by dpgp_trsm_batched on the identity, the M x M products as plain
    fp64 GEMMs (the library's strided MFMA kernel, ``matmul`` above), element-wise work in torch.  Reads Psi2, Psi1^T y and y^T y from the
    workspace of the forward evaluation (dpgp_elbo_workspace_layout); K_uu is rebuilt (one gram launch).
        G_B = -1/2 B^-1 - 1/2 beta^2 w w^T,  w = B^-1 v;   G_K = 1/2 K^-1 - 1/2 beta K^-1 P K^-1 + G_B;
        G_P = 1/2 beta K^-1 + beta G_B;   G_v = beta^2 w;   d/dalpha, d/dbeta complete (see chain_grad_kernel)."""
    import ctypes
    f64 = torch.float64
    d, n, m, q = workspace.shape
    dev = workspace.ws.device
    off_p2, ns2, esz, mp, off_v, ns1, off_yy, nyy = workspace.layout[:8]
    ns2 = ns2 if psi2_slabs is None else int(psi2_slabs)        # (after elbo_fhat_step: one slab)
    raw = workspace.ws
    pdt = torch.float32 if esz == 4 else f64
    p2 = raw[off_p2:off_p2 + esz * ns2 * d * mp * mp].view(pdt).view(ns2, d, mp, mp).sum(dim=0, dtype=f64)[:, :m, :m]
    p2 = torch.tril(p2) + torch.tril(p2, -1).transpose(1, 2)                    # (lower patches are what the kernel writes)
    v = raw[off_v:off_v + 8 * ns1 * d * m].view(f64).view(ns1, d, m).sum(dim=0)
    yy = raw[off_yy:off_yy + 8 * nyy * d].view(f64).view(nyy, d).sum(dim=0)
    k_uu = ard_rbf_gram(_prep(z, f64, 'z'), None, gamma, alpha, beta, include_noise=False, include_jitter=True, jitter=jitter)
    l_k, info_k = potrf_batched(k_uu)
    l_b, info_b = potrf_batched(k_uu + beta[:, None, None] * p2)
    li = tril_inverse_batched(l_k)
    k_inv = matmul(li.transpose(1, 2), li)
    li = tril_inverse_batched(l_b)
    b_inv = matmul(li.transpose(1, 2), li)
    del li
    w = matmul(b_inv, v[:, :, None])[:, :, 0]
    vw = torch.sum(v * w, dim=1)
    x = matmul(matmul(k_inv, p2), k_inv)
    be = beta[:, None, None]
    gb = -0.5 * b_inv - 0.5 * be * be * w[:, :, None] * w[:, None, :]
    gk = 0.5 * k_inv - 0.5 * be * x + gb
    gp_ = 0.5 * be * k_inv + be * gb
    wk_ = gk * (k_uu - float(jitter) * torch.eye(m, dtype=f64, device=dev))
    s_k, s_p = wk_.sum(dim=(1, 2)), (gp_ * p2).sum(dim=(1, 2))
    s_gbp, tr = (gb * p2).sum(dim=(1, 2)), (k_inv * p2).sum(dim=(1, 2))
    dab = torch.stack([-0.5 * beta * n + (s_k + 2.0 * s_p + beta * beta * vw) / alpha,
                       0.5 * n / beta + 0.5 * (tr - alpha * n) - 0.5 * yy + beta * vw + s_gbp], dim=1).contiguous()
    pad = (0, mp - m, 0, mp - m)
    gp = torch.nn.functional.pad(gp_, pad).contiguous()
    wk = torch.nn.functional.pad(wk_, pad).contiguous()
    gv = torch.nn.functional.pad(beta[:, None] ** 2 * w, (0, mp - m)).contiguous()
    return gp, wk, gv, dab, torch.maximum(info_k, info_b)


def elbo_grad_psi(y, z, mu, s, gamma, alpha, g_psi2, w_kuu, g_v, prec='mixed', g_psi1=None):
    """Backward pass, stage B: d f_hat / d (mu [N,Q], s [N,Q], z [M,Q], gamma [D,Q]) from the stage-A adjoints
    (rbf_kernel.py:58-199 differentiated).  g_psi1 [D,N,Mp] (mixed precision only): a full adjoint of Psi1 in place of the
    rank-1 form g_v[d,a] y[n,d] — then y and g_v may be None."""
    f64 = torch.float64
    z, mu, s = _prep(z, f64, 'z'), _prep(mu, f64, 'mu'), _prep(s, f64, 's')
    gamma, alpha, _, d = _hyp(gamma, alpha, None, f64)
    n, m, q = mu.shape[0], z.shape[0], z.shape[1]
    if y is not None and (y.dtype != f64 or y.stride(1) != 1):
        y = y.to(f64).contiguous()
    if g_psi1 is not None:
        g_psi1 = _prep(g_psi1, f64, 'g_psi1')
        assert g_psi1.shape == (d, n, g_psi2.shape[1])
    dev = mu.device
    if prec == 'f64' and m > 128:
        if g_psi1 is not None:
            raise ValueError('a full Psi1 adjoint is taken in mixed precision only')
        return _elbo_grad_psi_f64_blocks(y, z, mu, s, gamma, alpha, g_psi2, w_kuu, g_v)
    l = _lib.lib()
    wsb = l.dpgp_elbo_grad_psi_workspace_bytes_ex(d, n, m, q, _lib.PREC[prec])
    ws = _ws(wsb, dev)
    dmu, ds = torch.empty((n, q), dtype=f64, device=dev), torch.empty((n, q), dtype=f64, device=dev)
    dz, dg = torch.empty((m, q), dtype=f64, device=dev), torch.empty((d, q), dtype=f64, device=dev)
    _lib.check(l.dpgp_elbo_grad_psi_ex(d, n, m, q, y.data_ptr() if y is not None else None, y.stride(0) if y is not None else 0,
                                       z.data_ptr(), mu.data_ptr(), s.data_ptr(), gamma.data_ptr(), alpha.data_ptr(),
                                       g_psi2.data_ptr(), w_kuu.data_ptr(), g_v.data_ptr() if g_v is not None else None,
                                       g_psi1.data_ptr() if g_psi1 is not None else None, _lib.PREC[prec], ws.data_ptr(), wsb,
                                       dmu.data_ptr(), ds.data_ptr(), dz.data_ptr(), dg.data_ptr(), _stream()),
               'dpgp_elbo_grad_psi_ex')
    return dmu, ds, dz, dg


def _elbo_grad_psi_f64_blocks(y, z, mu, s, gamma, alpha, g_psi2, w_kuu, g_v):
    """Stage B in fp64 for M > 128.  The fp64 kernel keeps the M x M adjoint of one output dim in LDS (M <= 128); every term
    of stage B is a sum over PAIRS of inducing points (Psi2 and K_uu terms) or over single ones (Psi1 term), so the sum is
    split over the nb (nb - 1) / 2 unordered pairs {I, J} of blocks of <= 64 inducing points: pair {I, J} runs the same
    kernel on the 128 points z_I, z_J with the adjoint blocks [[G_II / (nb - 1), G_IJ], [G_JI, G_JJ / (nb - 1)]] (a diagonal
    block and a single point belong to nb - 1 pairs).  Exact in exact arithmetic, fp64 throughout, the same work as one
    pass over the M x M square; the training configuration (precision='f64', backward_precision='mixed') does not come
    here — this is the reference-precision check of it."""
    f64 = torch.float64
    dev = mu.device
    d, (n, q), m = gamma.shape[0], mu.shape, z.shape[0]
    nb = -(-m // 64)
    blocks = [b.to(dev) for b in torch.tensor_split(torch.arange(m), nb)]
    wd = 1.0 / (nb - 1)
    dmu, ds = torch.zeros((n, q), dtype=f64, device=dev), torch.zeros((n, q), dtype=f64, device=dev)
    dz, dg = torch.zeros((m, q), dtype=f64, device=dev), torch.zeros((d, q), dtype=f64, device=dev)
    for i in range(nb):
        for j in range(i + 1, nb):
            idx = torch.cat([blocks[i], blocks[j]])
            ni, ms = blocks[i].numel(), idx.numel()
            mps = 16 * ((ms + 15) // 16)
            scale = torch.ones((ms, ms), dtype=f64, device=dev)
            scale[:ni, :ni] = wd
            scale[ni:, ni:] = wd
            pad2 = (0, mps - ms, 0, mps - ms)
            gs = torch.nn.functional.pad(g_psi2[:, idx][:, :, idx] * scale, pad2).contiguous()
            ws_ = torch.nn.functional.pad(w_kuu[:, idx][:, :, idx] * scale, pad2).contiguous()
            gvs = torch.nn.functional.pad(g_v[:, idx] * wd, (0, mps - ms)).contiguous() if g_v is not None else None
            a, b, c, e = elbo_grad_psi(y, z[idx].contiguous(), mu, s, gamma, alpha, gs, ws_, gvs, prec='f64')
            dmu += a
            ds += b
            dz.index_add_(0, idx, c)
            dg += e
    return dmu, ds, dz, dg
