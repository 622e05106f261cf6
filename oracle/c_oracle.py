"""
TEST INFRASTRUCTURE — ctypes front-end of oracle/dpgp_oracle.c (see that file's header).  Imported only by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PD = ctypes.POINTER(ctypes.c_double)
_PI = ctypes.POINTER(ctypes.c_int)
FLAG_NOISE, FLAG_JITTER = 1, 2


def build():
    subprocess.check_call(['make', '-s', '-C', _HERE, 'all'])


def _p(a):
    return None if a is None else a.ctypes.data_as(_PD)


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class COracle:
    """fast=False: strict-IEEE build (the checker).  fast=True: -O3 -ffast-math build (the timed CPU baseline)."""

    def __init__(self, fast=False):
        name = 'libdpgp_oracle.so'
        if fast:                       # AVX-512 build when the host has it (8-wide libmvec exp), AVX2 otherwise
            try:
                flags = open('/proc/cpuinfo').read()
            except OSError:
                flags = ''
            name = 'libdpgp_oracle_fast_v4.so' if ' avx512f' in flags else 'libdpgp_oracle_fast.so'
        self.build_name = name
        path = os.path.join(_HERE, '_build', name)
        if not os.path.exists(path):
            build()
        self.lib = ctypes.CDLL(path)
        self.lib.dpgp_ref_kl_qx.restype = ctypes.c_double
        self.max_threads = int(self.lib.dpgp_ref_max_threads())

    def gram(self, x0, x1, gamma, alpha, beta, include_noise=False, include_jitter=False, jitter=1e-8, nthreads=1):
        x0 = _f(x0); gamma = _f(np.atleast_2d(gamma)); alpha = _f(alpha).reshape(-1); beta = _f(beta).reshape(-1)
        x1c = None if x1 is None else _f(x1)
        b, q = gamma.shape
        n0 = x0.shape[0]
        n1 = n0 if x1 is None else x1c.shape[0]
        out = np.empty((b, n0, n1))
        self.lib.dpgp_ref_gram(b, n0, n1, q, _p(x0), _p(x1c), _p(gamma), _p(alpha), _p(beta),
                               (FLAG_NOISE if include_noise else 0) | (FLAG_JITTER if include_jitter else 0),
                               ctypes.c_double(jitter), _p(out), nthreads)
        return out

    def _zms(self, z, mu, s, gamma, alpha):
        z, mu, s = _f(z), _f(mu), _f(s)
        gamma = _f(np.atleast_2d(gamma)); alpha = _f(alpha).reshape(-1)
        return z, mu, s, gamma, alpha, gamma.shape[0], mu.shape[0], z.shape[0], z.shape[1]

    def psi1(self, z, mu, s, gamma, alpha, nthreads=1):
        z, mu, s, gamma, alpha, b, n, m, q = self._zms(z, mu, s, gamma, alpha)
        out = np.empty((b, n, m))
        self.lib.dpgp_ref_psi1(b, n, m, q, _p(z), _p(mu), _p(s), _p(gamma), _p(alpha), _p(out), nthreads)
        return out

    def psi1T_y(self, z, mu, s, gamma, alpha, y, nthreads=1):
        z, mu, s, gamma, alpha, b, n, m, q = self._zms(z, mu, s, gamma, alpha)
        y = _f(y)
        out = np.empty((b, m))
        self.lib.dpgp_ref_psi1T_y(b, n, m, q, _p(z), _p(mu), _p(s), _p(gamma), _p(alpha), _p(y), y.shape[1], _p(out),
                                  nthreads)
        return out

    def psi2(self, z, mu, s, gamma, alpha, nthreads=1, literal=False):
        z, mu, s, gamma, alpha, b, n, m, q = self._zms(z, mu, s, gamma, alpha)
        out = np.empty((b, m, m))
        fn = self.lib.dpgp_ref_psi2_literal if literal else self.lib.dpgp_ref_psi2
        fn(b, n, m, q, _p(z), _p(mu), _p(s), _p(gamma), _p(alpha), _p(out), nthreads)
        return out

    def potrf(self, a):
        a = _f(a).copy()
        info = self.lib.dpgp_ref_potrf(a.shape[0], _p(a))
        return a, int(info)

    def trsm(self, l, rhs):
        l = _f(l); rhs = _f(rhs).copy()
        r2 = rhs.reshape(rhs.shape[0], -1)
        self.lib.dpgp_ref_trsm(l.shape[0], r2.shape[1], _p(l), _p(r2))
        return rhs

    def fhat_terms(self, y, z, mu, s, gamma, alpha, beta, jitter=1e-8, nthreads=1, return_parts=False):
        z, mu, s, gamma, alpha, d, n, m, q = self._zms(z, mu, s, gamma, alpha)
        y = _f(y); beta = _f(beta).reshape(-1)
        assert y.shape == (n, d)
        terms = np.empty((d, 5))
        info = np.zeros(d, dtype=np.int32)
        p2 = np.empty((d, m, m)) if return_parts else None
        kuu = np.empty((d, m, m)) if return_parts else None
        v = np.empty((d, m)) if return_parts else None
        self.lib.dpgp_ref_fhat_terms(d, n, m, q, _p(y), d, _p(z), _p(mu), _p(s), _p(gamma), _p(alpha), _p(beta),
                                     ctypes.c_double(jitter), _p(terms), info.ctypes.data_as(_PI), _p(p2), _p(kuu),
                                     _p(v), nthreads)
        if return_parts:
            return terms, info, dict(psi_2=p2, k_uu=kuu, psi1T_y=v)
        return terms, info

    def kl_qx(self, mu, s):
        mu, s = _f(mu), _f(s)
        return float(self.lib.dpgp_ref_kl_qx(mu.shape[0], mu.shape[1], _p(mu), _p(s)))
