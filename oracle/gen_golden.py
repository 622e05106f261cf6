"""
TEST INFRASTRUCTURE, CONTAINER-ONLY: golden-vector generator.  Run as ``python oracle/gen_golden.py`` in the build
container (needs /root/reference; the GPU box never has it and never runs this).

What it does, in order:
 1. puts ``oracle/standin`` (eager NumPy module named ``tensorflow``) and a symlink dir named ``dp_gp_lvm`` ->
    /root/reference on sys.path (the name satisfies src/utils/constants.py:76), with bytecode writing disabled so
    nothing is written into the read-only reference tree;
 2. RUNS THE REFERENCE'S OWN UNIT TESTS under that stand-in (kernel_unittests, dp_unittests, bgplvm objective,
    dpgplvm TestDPGPLVM/TestT1/TestD2T1.test_objective) — the reference's known-answer checks of its TF code
    against its pure-NumPy re-derivations — and refuses to write fixtures unless they all pass;
 3. evaluates the reference's source on the unit-test inputs and on the SURVEY §8d synthetic problems, evaluates
    ``oracle/dpgp_oracle.py`` (and the C restatement when built) on the same inputs, asserts agreement
    (rtol 1e-10), and writes inputs + reference outputs to ``tests/golden/*.npz``.

Fixtures are data only (inputs, expected outputs); no reference source text is stored.
"""

import os
import sys
import io
import time
import unittest

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
REF = '/root/reference'
LINK_DIR = '/tmp/dpgp_reflink'
os.makedirs(LINK_DIR, exist_ok=True)
LINK = os.path.join(LINK_DIR, 'dp_gp_lvm')
if not os.path.islink(LINK):
    os.symlink(REF, LINK)
sys.path[:0] = [os.path.join(HERE, 'standin'), LINK, REPO]

import numpy as np                                                           # noqa: E402
import matplotlib                                                            # noqa: E402
matplotlib.use('Agg')

import tensorflow as tf                                                      # noqa: E402  (the stand-in)
assert 'standin' in tf.__file__
from src.kernels.rbf_kernel import k_ard_rbf                                 # noqa: E402  (reference source)
from src.models.dp_gp_lvm import dp_gp_lvm                                   # noqa: E402
from src.models.dirichlet_process import dirichlet_process                   # noqa: E402
from src.models.expressions.gp_expressions import calculate_kl_divergence_standard_prior as ref_kl   # noqa: E402
from src.distributions.log_normal import log_pdf as ref_lognormal            # noqa: E402
import test.unittests.kernel_unittests as ku                                 # noqa: E402
import test.unittests.dp_unittests as du                                     # noqa: E402
import test.unittests.bgplvm_unittests as bu                                 # noqa: E402
import test.unittests.dpgplvm_unitttests as dgu                              # noqa: E402

from oracle import dpgp_oracle as orc                                        # noqa: E402
from dp_gp_lvm_amd.utils.synthetic import make_problem, CONFIGS              # noqa: E402

OUT = os.path.join(REPO, 'tests', 'golden')
os.makedirs(OUT, exist_ok=True)
RTOL = 1e-10


def close(a, b, what, rtol=RTOL, atol=0.0):
    np.testing.assert_allclose(np.asarray(a), np.asarray(b), rtol=rtol, atol=atol, err_msg=what)


def diag3(s):
    """[N,Q] variances -> [N,Q,Q] covariance, the form the reference API takes."""
    return tf.matrix_diag(s)


# --------------------------------------------------------------------------------------------------------------
# 2. the reference's own unit tests, run under the stand-in
# --------------------------------------------------------------------------------------------------------------

def run_reference_unit_tests():
    suite = unittest.TestSuite()
    ld = unittest.defaultTestLoader
    suite.addTests(ld.loadTestsFromTestCase(ku.TestRbfKernel))
    suite.addTests(ld.loadTestsFromTestCase(ku.TestRbfBatchKernel))
    suite.addTests(ld.loadTestsFromTestCase(du.TestDP))
    suite.addTest(bu.TestBGPLVM('test_objective'))
    # TestD1T1 (D=1,Q=1) cannot pass in the reference itself: its setUp trips the reference's own PCA assertion
    # 0 < Q < min(N, D) (src/utils/expressions.py:61) before any TensorFlow op runs, so it is not run here.
    for cls in (dgu.TestDPGPLVM, dgu.TestT1, dgu.TestD2T1):
        suite.addTest(cls('test_objective'))
    buf = io.StringIO()
    res = unittest.TextTestRunner(stream=buf, verbosity=1).run(suite)
    print('reference unit tests under the stand-in: ran %d, failures %d, errors %d'
          % (res.testsRun, len(res.failures), len(res.errors)))
    if not res.wasSuccessful():
        print(buf.getvalue())
        raise SystemExit('reference unit tests do not pass under the stand-in; refusing to write fixtures')
    return res.testsRun


# --------------------------------------------------------------------------------------------------------------
# 3a. kernel fixtures: inputs of TestRbfKernel / TestRbfBatchKernel setUp
# --------------------------------------------------------------------------------------------------------------

def kernel_fixture(case_cls, name):
    tc = case_cls('test_psi_2')
    tc.setUp()
    b = 1 if np.ndim(tc.gamma) == 1 else tc.gamma.shape[0]
    gamma = np.reshape(tc.gamma, (b, -1))
    alpha = np.reshape(tc.alpha, (b, 1))
    beta = np.reshape(tc.beta, (b, 1))
    kern = tc.kernel
    out = dict(x0=tc.x0, x1=tc.x1, x_mean=tc.x_mean, x_var=tc.x_var, x_u=tc.x_u,
               gamma=gamma, alpha=alpha, beta=beta)
    # gram: every (input_1, noise, jitter) combination the reference tests exercise (kernel_unittests.py:194-309)
    combos = [('xx', tc.x0, None), ('x01', tc.x0, tc.x1), ('x10', tc.x1, tc.x0), ('uu_same', tc.x_u, tc.x_u),
              ('uu', tc.x_u, None)]
    for tag, a, c in combos:
        for noise in (False, True):
            for jit in (False, True):
                ref = kern.covariance_matrix(a, c, include_noise=noise, include_jitter=jit)
                mine = orc.ard_rbf_gram(a, c, gamma, alpha, beta, noise, jit)
                close(mine, ref, 'gram %s %s %s' % (tag, noise, jit))
                # the reference's naive triple loop (B=1 per call)
                for bi in range(b):
                    naive = ku.k_ard_rbf_covariance_matrix_naive(a, gamma[bi], alpha[bi, 0], beta[bi, 0], input_1=c,
                                                                 include_noise=noise, include_jitter=jit)
                    close(ref[bi], naive, 'gram naive %s' % tag, rtol=1e-9)
                out['gram_%s_n%d_j%d' % (tag, noise, jit)] = ref
    for noise in (False, True):
        for jit in (False, True):
            ref = kern.covariance_diag(tc.x0, include_noise=noise, include_jitter=jit)
            close(orc.ard_rbf_diag(tc.x0.shape[0], alpha, beta, noise, jit), ref, 'diag')
            out['diag_n%d_j%d' % (noise, jit)] = np.asarray(ref)
    cov = diag3(tc.x_var)
    p0 = kern.psi_0(tc.x_u, tc.x_mean, cov)
    p1 = kern.psi_1(tc.x_u, tc.x_mean, cov)
    p2 = kern.psi_2(tc.x_u, tc.x_mean, cov)
    close(orc.psi0(tc.x_mean.shape[0], alpha), p0, 'psi0')
    close(orc.psi1(tc.x_u, tc.x_mean, tc.x_var, gamma, alpha), p1, 'psi1')
    close(orc.psi2(tc.x_u, tc.x_mean, tc.x_var, gamma, alpha), p2, 'psi2')
    for bi in range(b):
        close(p1[bi], ku.k_ard_rbf_psi_1_naive(tc.x_mean, tc.x_var, tc.x_u, gamma[bi], alpha[bi, 0]), 'psi1 naive',
              rtol=1e-9)
        if bi < 2:   # the naive psi2 is a 5-deep Python loop: two batch entries are enough to pin the formula
            close(p2[bi], ku.k_ard_rbf_psi_2_naive(tc.x_mean, tc.x_var, tc.x_u, gamma[bi], alpha[bi, 0]),
                  'psi2 naive', rtol=1e-9)
    out.update(psi_0=np.asarray(p0), psi_1=p1, psi_2=p2)
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **out)
    print('wrote', name, 'B=%d' % b)


# --------------------------------------------------------------------------------------------------------------
# 3b. model fixtures: the reference model built exactly as its unit tests build it
# --------------------------------------------------------------------------------------------------------------

def ref_fhat_per_d(y, mu, s, z, gamma, alpha, beta):
    """Per-output f_hat from the reference's own 'stable' expression (bgplvm_unittests.py:55-120, D=1 per call), with
    its psi/gram inputs taken from the reference kernel object instead of the 5-deep naive Python loops."""
    saved = (bu.k_ard_rbf_covariance_matrix_naive, bu.k_ard_rbf_psi_0_naive, bu.k_ard_rbf_psi_1_naive,
             bu.k_ard_rbf_psi_2_naive)

    def _kern(gam, alp, bet=1.0):
        return k_ard_rbf(gamma=np.reshape(gam, (1, -1)), alpha=np.reshape(alp, (1, 1)), beta=np.reshape(bet, (1, 1)))

    bu.k_ard_rbf_covariance_matrix_naive = lambda input_0, gamma, alpha, beta, input_1=None, include_noise=False, \
        include_jitter=False: _kern(gamma, alpha, beta).covariance_matrix(
            input_0, input_1, include_noise=include_noise, include_jitter=include_jitter)[0]
    bu.k_ard_rbf_psi_0_naive = lambda num_samples, alpha: num_samples * alpha
    bu.k_ard_rbf_psi_1_naive = lambda x_mean, x_var, x_u, gamma, alpha: ref_psi1_chunked(
        _kern(gamma, alpha), x_u, x_mean, x_var)[0]
    bu.k_ard_rbf_psi_2_naive = lambda x_mean, x_var, x_u, gamma, alpha: ref_psi2_chunked(
        _kern(gamma, alpha), x_u, x_mean, x_var)[0]
    try:
        out = np.array([float(np.squeeze(bu.free_energy_stable(y=y[:, d:d + 1], x_mean=mu, x_var=s, x_u=z,
                                                               gamma=gamma[d], alpha=alpha[d], beta=beta[d])))
                        for d in range(y.shape[1])])
    finally:
        (bu.k_ard_rbf_covariance_matrix_naive, bu.k_ard_rbf_psi_0_naive, bu.k_ard_rbf_psi_1_naive,
         bu.k_ard_rbf_psi_2_naive) = saved
    return out


def ref_psi2_chunked(kern, z, mu, s, budget=1.0e9):
    """Reference Kernel.psi_2 summed over N-chunks (additive over n, rbf_kernel.py:199) so that its [B,c,M,M,Q]
    temporary stays below ``budget`` bytes."""
    m, q = z.shape
    b = np.shape(kern.hyperparameters[list(kern.hyperparameters.keys())[0]])[0]
    c = max(1, int(budget / (8.0 * b * m * m * q)))
    acc = 0.0
    for n0 in range(0, mu.shape[0], c):
        acc = acc + kern.psi_2(z, mu[n0:n0 + c], diag3(s[n0:n0 + c]))
    return acc


def ref_psi1_chunked(kern, z, mu, s, budget=1.0e9):
    m, q = z.shape
    b = np.shape(kern.hyperparameters[list(kern.hyperparameters.keys())[0]])[0]
    c = max(1, int(budget / (8.0 * b * m * q)))
    return np.concatenate([kern.psi_1(z, mu[n0:n0 + c], diag3(s[n0:n0 + c])) for n0 in range(0, mu.shape[0], c)],
                          axis=1)


def model_fixture(case_cls, name):
    tf.reset_default_graph()
    tc = case_cls('test_objective')
    tc.setUp()
    model = tc.dpgplvm
    y = tc.y
    mu, cov = model.q_x
    s = tf.matrix_diag_part(cov)
    z = np.asarray(model.inducing_input)
    phi = np.asarray(model.dp.q_z)
    g1, g2 = (np.asarray(a) for a in model.dp.q_v)
    w1, w2 = (float(a) for a in model.dp.q_alpha)
    gat, aat, bat = (np.asarray(a) for a in model.dp_atoms)
    gamma, alpha, beta = (np.asarray(a) for a in (model.ard_weights, model.signal_variance, model.noise_precision))
    kern = model.kernel
    k_uu = kern.covariance_matrix(z, None, include_noise=False, include_jitter=True)
    p1 = kern.psi_1(z, mu, cov)
    p2 = kern.psi_2(z, mu, cov)
    ref_obj = float(model.objective)
    ref_dp = float(model.dp.objective)
    ref_klv = float(ref_kl(mu, cov))
    ref_hp = float(sum(np.sum(ref_lognormal(a)) for a in (gat, aat, bat)))
    fhat_d = ref_fhat_per_d(y, np.asarray(mu), np.asarray(s), z, gamma, alpha, beta)
    # the reference's naive known-answer objective (dpgplvm_unitttests.py:78-126)
    dp_elbo = du.elbo_naive(phi=phi, gamma_1=g1, gamma_2=g2, w_1=w1, w_2=w2, s_1=tc.s_1, s_2=tc.s_2)
    gp_naive = -bu.kl_qx_px_naive(x_mean=np.asarray(mu), x_var=np.asarray(s))
    for d in range(y.shape[1]):
        gp_naive += float(np.squeeze(bu.free_energy_naive(y=y[:, d:d + 1], x_mean=np.asarray(mu), x_var=np.asarray(s),
                                                          x_u=z, gamma=gamma[d], alpha=alpha[d], beta=beta[d])))
    obj_naive = float(-(dp_elbo + gp_naive + ref_hp))
    close(obj_naive, ref_obj, 'naive known answer vs model', rtol=1e-7)

    # the oracle against all of it
    mg, ma, mb = orc.mix_hyperparameters(phi, gat, aat, bat)
    close(mg, gamma, 'mix gamma'); close(ma, alpha, 'mix alpha'); close(mb, beta, 'mix beta')
    terms, parts = orc.fhat_terms(y, z, mu, s, gamma, alpha, beta, return_parts=True)
    close(parts['k_uu'], k_uu, 'k_uu'); close(parts['psi_2'], p2, 'psi2')
    close(parts['psi1T_y'], np.einsum('dnm,nd->dm', p1, y), 'psi1T_y')
    close(terms.sum(axis=1), fhat_d, 'per-d f_hat', rtol=1e-9)
    close(orc.kl_qx(mu, s), ref_klv, 'kl')
    close(orc.hyperprior(gat, aat, bat), ref_hp, 'hyperprior')
    close(orc.dp_objective(phi, g1, g2, w1, w2, tc.s_1, tc.s_2), ref_dp, 'dp objective')
    mine = orc.objective(y, z, mu, s, phi, gat, aat, bat, g1, g2, w1, w2, tc.s_1, tc.s_2)
    close(mine, ref_obj, 'objective', rtol=1e-11)

    np.savez_compressed(
        os.path.join(OUT, name + '.npz'),
        y=y, mu=np.asarray(mu), s=np.asarray(s), z=z, phi=phi, g1=g1, g2=g2, w1=w1, w2=w2, s1=tc.s_1, s2=tc.s_2,
        gamma_atoms=gat, alpha_atoms=aat, beta_atoms=bat, gamma=gamma, alpha=alpha, beta=beta,
        k_uu=k_uu, psi_1=p1, psi_2=p2, l_uu=parts['l_uu'], l_a=parts['l_a'], fhat_per_d=fhat_d, fhat_terms=terms,
        kl=ref_klv, hyperprior=ref_hp, dp_objective=ref_dp, objective=ref_obj, objective_naive=obj_naive)
    print('wrote %s  objective %.12f (naive known answer %.12f)' % (name, ref_obj, obj_naive))


# --------------------------------------------------------------------------------------------------------------
# 3c. end-to-end fixtures on the synthetic recipe (small shapes: all of D; BASELINE shapes: 4 chosen d)
# --------------------------------------------------------------------------------------------------------------

def ref_model_pieces(p, dsel):
    """Reference kernel object (B = len(dsel)) evaluated on problem p, chunked over n."""
    gamma, alpha, beta = p['gamma'][dsel], p['alpha'][dsel], p['beta'][dsel]
    kern = k_ard_rbf(gamma=gamma, alpha=alpha, beta=beta)
    k_uu = kern.covariance_matrix(p['z'], None, include_noise=False, include_jitter=True)
    p2 = ref_psi2_chunked(kern, p['z'], p['mu'], p['s'])
    p1 = ref_psi1_chunked(kern, p['z'], p['mu'], p['s'])
    v = np.einsum('dnm,nd->dm', p1, p['y'][:, dsel])
    return k_uu, p1, p2, v


def ref_fhat_from_pieces(y, k_uu, p1ty, p2, alpha, beta):
    """dp_gp_lvm.py:113-145 through the stand-in's tf ops (cholesky / triangular solves as the reference orders them),
    with the two [D,M,N] solves applied to Psi1^T y_d — the [D,N,N] product of :134 is never formed."""
    n = y.shape[0]
    d, m = p1ty.shape
    beta_d11 = tf.expand_dims(beta, axis=-1)
    l_uu = tf.cholesky(k_uu)
    x = tf.matrix_triangular_solve(l_uu, p2, lower=True)
    t2 = tf.transpose(tf.matrix_triangular_solve(l_uu, tf.transpose(x, perm=[0, 2, 1]), lower=True), perm=[0, 2, 1])
    a = beta_d11 * t2 + tf.eye(m, batch_shape=[d], dtype=tf.float64)
    l_a = tf.cholesky(a)
    cy = tf.matrix_triangular_solve(l_a, tf.matrix_triangular_solve(l_uu, p1ty[:, :, None], lower=True), lower=True)
    return (0.5 * n * (np.log(beta[:, 0]) - np.log(2.0 * np.pi))
            - np.sum(np.log(tf.matrix_diag_part(l_a)), axis=-1)
            + 0.5 * beta[:, 0] * (tf.trace(t2) - alpha[:, 0] * n)
            - 0.5 * beta[:, 0] * np.sum(y * y, axis=0)
            + 0.5 * beta[:, 0] ** 2 * np.sum(cy[:, :, 0] ** 2, axis=-1)), l_uu, l_a


def synthetic_fixture(cfg, name, dsel=None, full_reference_model=False):
    t0 = time.time()
    p = make_problem(cfg)
    n, d, m, q = CONFIGS[cfg]
    dsel = np.arange(d) if dsel is None else np.asarray(dsel)
    k_uu, p1, p2, v = ref_model_pieces(p, dsel)
    fhat_d, l_uu, l_a = ref_fhat_from_pieces(p['y'][:, dsel], k_uu, v, p2, p['alpha'][dsel], p['beta'][dsel])
    out = dict(cfg=cfg, dsel=dsel, fhat_per_d=fhat_d, psi1T_y=v,
               kl=float(ref_kl(p['mu'], diag3(p['s']))),
               hyperprior=float(sum(np.sum(ref_lognormal(p[k])) for k in ('gamma_atoms', 'alpha_atoms', 'beta_atoms'))))
    # the dp objective from the reference's naive NumPy known-answer function
    out['dp_objective'] = float(-du.elbo_naive(phi=p['phi'], gamma_1=p['g1'], gamma_2=p['g2'], w_1=p['w1'], w_2=p['w2'],
                                               s_1=p['s1'], s_2=p['s2']))
    sub = dict(p, y=p['y'][:, dsel], gamma=p['gamma'][dsel], alpha=p['alpha'][dsel], beta=p['beta'][dsel])
    terms, parts = orc.fhat_terms(sub['y'], p['z'], p['mu'], p['s'], sub['gamma'], sub['alpha'], sub['beta'],
                                  return_parts=True)
    close(parts['k_uu'], k_uu, 'k_uu'); close(parts['psi_2'], p2, 'psi2'); close(parts['psi1T_y'], v, 'psi1T_y')
    close(terms.sum(axis=1), fhat_d, 'per-d f_hat', rtol=1e-9)
    close(orc.kl_qx(p['mu'], p['s']), out['kl'], 'kl')
    close(orc.hyperprior(p['gamma_atoms'], p['alpha_atoms'], p['beta_atoms']), out['hyperprior'], 'hyperprior')
    close(orc.dp_objective(p['phi'], p['g1'], p['g2'], p['w1'], p['w2'], p['s1'], p['s2']), out['dp_objective'], 'dp')
    out['fhat_terms'] = terms
    if len(dsel) == d:
        # small shapes: everything, including the literal reference f_hat expression with its [D,N,N] product
        out.update(k_uu=k_uu, psi_1=p1, psi_2=p2, l_uu=l_uu, l_a=l_a)
        out['objective'] = out['dp_objective'] - (fhat_d.sum() - out['kl']) - out['hyperprior']
        mine = orc.objective(p['y'], p['z'], p['mu'], p['s'], p['phi'], p['gamma_atoms'], p['alpha_atoms'],
                             p['beta_atoms'], p['g1'], p['g2'], p['w1'], p['w2'], p['s1'], p['s2'])
        close(mine, out['objective'], 'objective', rtol=1e-11)
        lit = literal_reference_fhat(p)
        close(lit, fhat_d.sum(), 'literal dp_gp_lvm.py:108-145 f_hat', rtol=1e-10)
        out['fhat_literal'] = lit
    else:
        # BASELINE shapes: checksums of the [M,M] matrices + 16 sampled entries per selected d
        rs = np.random.default_rng(7)
        ii, jj = rs.integers(0, m, 16), rs.integers(0, m, 16)
        out.update(psi_2_fro=np.sqrt(np.sum(p2 * p2, axis=(1, 2))), psi_2_samples=p2[:, ii, jj], sample_i=ii,
                   sample_j=jj, k_uu_fro=np.sqrt(np.sum(k_uu * k_uu, axis=(1, 2))), k_uu_samples=k_uu[:, ii, jj],
                   psi_2_rowsum=p2.sum(axis=2), logdet_l_a=np.sum(np.log(tf.matrix_diag_part(l_a)), axis=-1))
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **out)
    print('wrote %s (N,D,M,Q)=%s  d=%s  %.1fs' % (name, CONFIGS[cfg], list(dsel[:6]), time.time() - t0))


def literal_reference_fhat(p):
    """The f_hat block of dp_gp_lvm.py:108-145 re-evaluated op for op on the reference kernel object, including the
    [D,M,N] solves and the [D,N,N] product (small shapes only)."""
    y = p['y']
    n, d = y.shape
    m = p['z'].shape[0]
    beta = p['beta']
    kern = k_ard_rbf(gamma=p['gamma'], alpha=p['alpha'], beta=beta)
    cov = diag3(p['s'])
    psi_0 = kern.psi_0(p['z'], p['mu'], cov)
    psi_1 = kern.psi_1(p['z'], p['mu'], cov)
    psi_2 = kern.psi_2(p['z'], p['mu'], cov)
    beta_d11 = tf.expand_dims(beta, axis=-1)
    k_uu = kern.covariance_matrix(p['z'], None, include_noise=False, include_jitter=True)
    l_uu = tf.cholesky(k_uu)
    x = tf.matrix_triangular_solve(l_uu, psi_2, lower=True)
    t2 = tf.transpose(tf.matrix_triangular_solve(l_uu, tf.transpose(x, perm=[0, 2, 1]), lower=True), perm=[0, 2, 1])
    a = beta_d11 * t2 + tf.eye(m, batch_shape=[d], dtype=tf.float64)
    l_a = tf.cholesky(a)
    log_det_l_a = tf.reduce_sum(tf.log(tf.matrix_diag_part(l_a)))
    c = tf.matrix_triangular_solve(l_a, tf.matrix_triangular_solve(l_uu, tf.transpose(psi_1, perm=[0, 2, 1]),
                                                                   lower=True), lower=True)
    ctc = tf.matmul(c, c, transpose_a=True)
    yb = tf.expand_dims(tf.transpose(y) * beta, axis=1)
    return float(0.5 * n * (tf.reduce_sum(tf.log(beta)) - d * np.log(2.0 * np.pi)) - log_det_l_a
                 + 0.5 * tf.reduce_sum(beta * (tf.reduce_sum(tf.matrix_diag_part(t2), axis=-1, keepdims=True) - psi_0))
                 - 0.5 * tf.reduce_sum(beta * tf.expand_dims(tf.diag_part(tf.matmul(y, y, transpose_a=True)), axis=-1))
                 + 0.5 * tf.reduce_sum(tf.matmul(tf.matmul(yb, ctc), yb, transpose_b=True)))


if __name__ == '__main__':
    only = set(sys.argv[1:])
    n_tests = run_reference_unit_tests()
    if not only or 'kernel' in only:
        kernel_fixture(ku.TestRbfKernel, 'kernel_b1')
        kernel_fixture(ku.TestRbfBatchKernel, 'kernel_b7')
    if not only or 'model' in only:
        model_fixture(dgu.TestDPGPLVM, 'dpgplvm_50_10_25_3_T8')
        model_fixture(dgu.TestT1, 'dpgplvm_T1_d5')
        model_fixture(dgu.TestD2T1, 'dpgplvm_d2')
    if not only or 'small' in only:
        synthetic_fixture(1, 'plumbing_100_12_20_4')
        synthetic_fixture(6, 'script_100_20_25_10')
    if not only or 'spot' in only:
        synthetic_fixture(2, 'spot_C2', dsel=[0, 21, 42, 63])
        synthetic_fixture(3, 'spot_C3', dsel=[0, 170, 341, 511])
        synthetic_fixture(5, 'spot_C5', dsel=[0, 186, 373, 559])
    if 'spot4' in only:
        synthetic_fixture(4, 'spot_C4', dsel=[0, 85, 170, 255])
    print('done; reference unit tests passed under the stand-in: %d' % n_tests)
