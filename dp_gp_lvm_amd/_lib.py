"""
ctypes binding of libdpgp_hip.so (the C ABI declared in include/dpgp.h).

The HIP library is the product: there is no CPU or PyTorch fallback.  If the shared object has not been built
(``python -c "import __graft_entry__ as g; g.build()"`` or ``make -C dp_gp_lvm_amd/csrc``) every operator raises.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# DPGP_LIBRARY overrides the in-tree path (used to load diagnostic builds of the same ABI)
LIB_PATH = os.environ.get('DPGP_LIBRARY') or os.path.join(_HERE, 'csrc', 'libdpgp_hip.so')

FLAG_NOISE, FLAG_JITTER = 1, 2
ALGO = {'auto': 0, 'plain': 1, 'mfma_f32': 2, 'patch_f16': 3}
PREC = {'f32': 0, 'mixed': 1, 'f64': 2, 'mixed_patch': 3, 'mixed_fast': 4}

_vp, _i, _d, _sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_double, ctypes.c_size_t
_ll = ctypes.c_longlong

# name -> (restype, argtypes); one entry per symbol declared in include/dpgp.h
SIGNATURES = {
    'dpgp_version': (_i, []),
    'dpgp_last_hip_error': (ctypes.c_char_p, []),
    'dpgp_psi1T_y_workspace_bytes': (_sz, [_i, _i, _i]),
    'dpgp_psi2_workspace_bytes': (_sz, [_i, _i, _i, _i, _i]),
    'dpgp_potrf_workspace_bytes': (_sz, [_i, _i, _i]),
    'dpgp_trsm_workspace_bytes': (_sz, [_i, _i, _i, _i]),
    'dpgp_elbo_workspace_bytes': (_sz, [_i, _i, _i, _i, _i]),
    'dpgp_elbo_workspace_layout': (_i, [_i, _i, _i, _i, _i, _vp]),
    'dpgp_elbo_fhat': (_i, [_i, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _d, _i, _i, _vp, _vp, _vp, _vp, _sz,
                            _vp]),
    'dpgp_elbo_fhat_ex': (_i, [_i, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _d, _i, _i, _vp, _vp, _vp, _vp, _sz,
                               _vp, _vp]),
    'dpgp_elbo_fhat_t_workspace_bytes': (_sz, [_i, _i, _i, _i, _i, _i]),
    'dpgp_elbo_fhat_t': (_i, [_i, _i, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _ll, _ll, _d, _i, _vp, _vp,
                              _vp, _vp, _vp, _sz, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    'dpgp_elbo_grad_chain': (_i, [_i, _i, _i, _i, _vp, _vp, _d, _i, _vp, _sz, _vp, _vp, _vp, _vp, _vp, _vp]),
    'dpgp_elbo_grad_psi_workspace_bytes': (_sz, [_i, _i, _i, _i]),
    'dpgp_elbo_grad_psi_workspace_bytes_ex': (_sz, [_i, _i, _i, _i, _i]),
    'dpgp_elbo_fhat_step': (_i, [_i, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _d, _vp, _vp, _vp, _vp, _sz, _vp, _sz, _vp, _vp]),
    'dpgp_elbo_grad_psi_step': (_i, [_i, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _sz, _vp, _sz, _vp, _vp,
                                     _vp, _vp, _vp]),
    'dpgp_elbo_step': (_i, [_i, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _d, _i, _vp, _vp, _vp, _vp, _sz, _vp, _vp, _vp, _vp,
                            _vp, _vp, _sz, _vp, _vp, _vp, _vp, _vp, _vp]),
    'dpgp_elbo_grad_psi': (_i, [_i, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _sz, _vp, _vp, _vp,
                                _vp, _vp]),
    'dpgp_elbo_grad_psi_ex': (_i, [_i, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _sz, _vp, _vp,
                                   _vp, _vp, _vp]),
    'dpgp_event_create': (_vp, []),
    'dpgp_event_destroy': (None, [_vp]),
    'dpgp_stream_create': (_vp, []),
    'dpgp_stream_destroy': (None, [_vp]),
    'dpgp_event_elapsed_ms': (ctypes.c_float, [_vp, _vp]),
    'dpgp_model_prepare': (_i, [_i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _d, _d, _i, _vp, _vp, _vp,
                                _vp, _vp, _vp, _vp]),
    'dpgp_model_prepare_t': (_i, [_i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _d, _d, _i, _vp, _vp, _vp,
                                  _vp, _vp]),
    'dpgp_model_scal_count': (_i, [_i]),
    'dpgp_model_backward': (_i, [_i] * 8 + [_vp] * 10 + [_d, _d, _i] + [_vp] * 15 + [_vp]),
    'dpgp_model_backward_t': (_i, [_i] * 8 + [_vp] * 10 + [_d, _d, _i] + [_vp] * 16 + [_vp]),
    'dpgp_model_pack': (_i, [_i, _vp, _vp, _vp, _vp]),
    'dpgp_model_finalize': (_i, [_vp, _vp, _vp, _vp, _vp]),
    'dpgp_elbo_grad_chain_big_workspace_bytes': (_sz, [_i, _i]),
    'dpgp_elbo_grad_chain_big': (_i, [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _d, _i, _vp, _sz, _i, _vp, _sz, _vp, _vp, _vp, _vp, _vp, _vp]),
    'dpgp_trouble_flag': (_i, [_sz, _vp, _i, _vp, _vp, _vp]),
    'dpgp_trtri_lower_batched_f64': (_i, [_i, _i, _vp, _vp, _vp, _sz, _vp]),
    'dpgp_gemm_strided_f64': (_i, [_i, _i, _i, _i, _d, _vp, _ll, _ll, _ll, _vp, _ll, _ll, _ll, _d, _vp, _ll, _ll, _ll, _vp]),
}
for _t in ('f32', 'f64'):
    SIGNATURES.update({
        'dpgp_ard_rbf_gram_' + _t: (_i, [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _d, _vp, _vp]),
        'dpgp_ard_rbf_diag_' + _t: (_i, [_i, _i, _vp, _vp, _i, _d, _vp, _vp]),
        'dpgp_psi0_' + _t: (_i, [_i, _i, _vp, _vp, _vp]),
        'dpgp_psi1_' + _t: (_i, [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
        'dpgp_psi1T_y_' + _t: (_i, [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _sz, _vp]),
        'dpgp_psi2_' + _t: (_i, [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _i, _vp]),
        'dpgp_potrf_batched_' + _t: (_i, [_i, _i, _vp, _vp, _vp, _sz, _i, _vp]),
        'dpgp_trsm_batched_' + _t: (_i, [_i, _i, _i, _vp, _vp, _vp, _sz, _i, _vp]),
        'dpgp_kl_qx_' + _t: (_i, [_i, _i, _vp, _vp, _vp, _vp]),
    })



class ExecResources(ctypes.Structure):
    """dpgp_exec_t of include/dpgp.h: optional psi2 timing events and model-level tail pointers."""
    _fields_ = [('ev_psi2_begin', _vp), ('ev_psi2_end', _vp), ('model_scal', _vp), ('model_pack', _vp),
                ('model_out', _vp), ('stream_aux', _vp), ('ev_fork', _vp), ('ev_join', _vp)]


_lib = None


class DpgpLibraryMissing(RuntimeError):
    pass


def lib():
    """The loaded library; raises loudly when it has not been built (no fallback path exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise DpgpLibraryMissing(
                'libdpgp_hip.so is not built: run `make -C %s` (hipcc, gfx950). dp_gp_lvm_amd has no CPU fallback.'
                % os.path.join(_HERE, 'csrc'))
        # torch ships its own libamdhip64 (same soname as /opt/rocm's): it must be the one already mapped when this
        # library's NEEDED entry is resolved, otherwise the process ends up with two HIP runtimes and no device.
        import torch  # noqa: F401
        l = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)          # AttributeError here = header/library mismatch
            fn.restype, fn.argtypes = res, args
        _lib = l
    return _lib


def check(rc, name):
    if rc != 0:
        if rc == -100:
            raise RuntimeError('%s: HIP kernel launch failed: %s' % (name, lib().dpgp_last_hip_error().decode()))
        raise ValueError('%s: bad argument #%d' % (name, -rc))
