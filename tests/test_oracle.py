# -*- coding: utf-8 -*-
"""Pins the CPU oracle (oracle/celerite_ref.c) before anything trusts it.

Sources of truth, all from the reference (paths relative to its checkout):
  * the printed known answer docs/tutorials/first.rst:101;
  * the dense-LAPACK identities of tests/test_celerite.py (log_determinant :45-85,
    solve :88-151, dot :154-192, dot_L :194-235, log_likelihood :311-404,
    nyquist :501-525) on their seeded inputs;
  * cpp/src/test_solvers.cc:13,79-97: |Cholesky - dense| <= 1e-10 on
    log_determinant and dot_solve, N = 1024, real / complex / mixed / general;
  * tests/golden/ipynb_golden.npz: outputs of the authors' NumPy prototype
    (cholesky.ipynb cells 0, 4, 5), an implementation independent of cholesky.h;
  * an mpmath (40-digit) dense LDL^T.
"""
import json
import os

import numpy as np
import pytest

from oracle import dense, ref
from _cases import (COEFFS_W4, COEFFS_W10, COEFFS_DOT, COEFFS_CC_REAL, COEFFS_CC_COMP, NO_GENERAL,
                    FIRST_TUTORIAL_LOGLIKE, first_tutorial_case, general_terms, logdet_case,
                    solve_case, synthetic, coeffs_of)

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def sho_complex(log_S0, log_Q, log_w0):
    """celerite/terms.py:503-517 written out (Q >= 1/2)."""
    S0, Q, w0 = np.exp(log_S0), np.exp(log_Q), np.exp(log_w0)
    f = np.sqrt(4.0 * Q ** 2 - 1)
    return S0 * w0 * Q, S0 * w0 * Q / f, 0.5 * w0 / Q, 0.5 * w0 / Q * f


def test_first_tutorial_known_answer():
    t, yerr, y = first_tutorial_case()
    Q, w0 = 1.0 / np.sqrt(2.0), 3.0
    k1 = sho_complex(np.log(np.var(y) / (w0 * Q)), np.log(Q), np.log(w0))
    Q = 1.0
    k2 = sho_complex(np.log(np.var(y) / (w0 * Q)), np.log(Q), np.log(w0))
    co = [np.array([[k1[i], k2[i]]]) for i in range(4)]
    ll, ld, q, st = ref.batch_log_likelihood(0.0, np.empty((1, 0)), np.empty((1, 0)), co[0], co[1],
                                             co[2], co[3], t, yerr ** 2, y - np.mean(y))
    assert st[0] == 0
    # the reference printed 16 significant digits; the oracle reproduces them all
    assert abs(ll[0] - FIRST_TUTORIAL_LOGLIKE) < 5e-14


@pytest.mark.parametrize("coeffs", [COEFFS_W4, COEFFS_W10])
def test_log_determinant_vs_dense(coeffs):
    t, diag = logdet_case()
    s = ref.RefSolver()
    s.compute(0.0, *coeffs, *NO_GENERAL, t, diag)
    K = dense.dense_matrix(0.0, *coeffs, *NO_GENERAL, t, diag)
    assert abs(s.log_determinant() - dense.dense_logdet(K)) < 1e-12


@pytest.mark.parametrize("with_general", [False, True])
@pytest.mark.parametrize("coeffs", [COEFFS_W4, COEFFS_W10])
def test_solve_vs_dense(coeffs, with_general):
    t, diag, b, gen = solve_case(with_general)
    s = ref.RefSolver()
    with pytest.raises(RuntimeError):
        s.log_determinant()
    with pytest.raises(RuntimeError):
        s.dot_solve(b)
    s.compute(0.0, *coeffs, *gen, t, diag)
    K = dense.dense_matrix(0.0, *coeffs, *gen, t, diag)
    x0 = np.linalg.solve(K, b)
    assert np.abs(s.solve(b)[:, 0] - x0).max() < 1e-9
    assert abs(s.dot_solve(b) - b @ x0) < 1e-9
    B5 = np.random.randn(len(t), 5)
    assert np.abs(s.solve(B5) - np.linalg.solve(K, B5)).max() < 1e-9
    assert abs(s.log_determinant() - dense.dense_logdet(K)) < 1e-9


@pytest.mark.parametrize("with_general", [False, True])
def test_dot_and_dot_L_vs_dense(with_general):
    np.random.seed(42)
    t = np.sort(np.random.rand(500))
    b = np.random.randn(len(t), 5)
    gen = general_terms(t, np.random.rand) if with_general else NO_GENERAL
    K = dense.dense_matrix(0.0, *COEFFS_DOT, *gen, t, np.zeros_like(t))
    s = ref.RefSolver()
    assert np.abs(s.dot(0.0, *COEFFS_DOT, *gen, t, b) - K @ b).max() < 1e-10

    np.random.seed(42)
    t = np.sort(np.random.rand(5))
    b = np.random.randn(len(t), 5)
    yerr = np.random.uniform(0.1, 0.5, len(t))
    gen = general_terms(t, np.random.rand) if with_general else NO_GENERAL
    K = dense.dense_matrix(0.0, *COEFFS_DOT, *gen, t, yerr ** 2)
    s.compute(0.0, *COEFFS_DOT, *gen, t, yerr ** 2)
    assert np.abs(s.dot_L(b) - np.linalg.cholesky(K) @ b).max() < 1e-12


def test_cc_suite_structure_1e10():
    """cpp/src/test_solvers.cc: four kernels, N = 1024, jitter 0.01, tolerance 1e-10 abs."""
    rng = np.random.RandomState(42)
    N = 1024
    x = np.sort(rng.uniform(-1, 1, N))
    yerr2 = 0.3 + 0.1 * rng.uniform(-1, 1, N)
    y = np.sin(x)
    U = np.array([x ** j for j in range(3)])
    V = np.array([x ** j / (1.0 + j) for j in range(3)])
    A = 1e-6 + np.sum(U * V, axis=0)
    e = np.empty(0)
    cases = [
        (COEFFS_CC_REAL + (e, e, e, e), NO_GENERAL),
        ((e, e) + COEFFS_CC_COMP, NO_GENERAL),
        (COEFFS_CC_REAL + COEFFS_CC_COMP, NO_GENERAL),
        (COEFFS_CC_REAL + COEFFS_CC_COMP, (A, U, V)),
    ]
    for coeffs, gen in cases:
        s = ref.RefSolver()
        s.compute(0.01, *coeffs, *gen, x, yerr2)
        K = dense.dense_matrix(0.01, *coeffs, *gen, x, yerr2)
        assert abs(s.log_determinant() - dense.dense_logdet(K)) < 1e-10
        assert abs(s.dot_solve(y) - y @ np.linalg.solve(K, y)) < 1e-10


def test_against_authors_numpy_prototype():
    g = np.load(os.path.join(GOLDEN, "ipynb_golden.npz"))
    a, b, c, d, t = g["a"], g["b"], g["c"], g["d"], g["t"]
    e = np.empty(0)
    diag = g["diag_full"] - np.sum(a)  # compute() adds sum(a_comp) itself
    s = ref.RefSolver()
    s.compute(0.0, e, e, a, b, c, d, *NO_GENERAL, t, diag)
    _, N, J, logdet, phi, u, W, D = s.state()
    assert np.allclose(D, g["D"], rtol=1e-11, atol=0)
    # the prototype keeps cos-like rows in X1 and sin-like rows in X2
    assert np.allclose(W[0::2].T, g["X1"], rtol=1e-9, atol=1e-13)
    assert np.allclose(W[1::2].T, g["X2"], rtol=1e-9, atol=1e-13)
    assert abs(logdet - float(g["logdet_chol"])) < 1e-10
    assert abs(logdet - float(g["logdet_dense"])) < 1e-9
    assert np.abs(s.solve(g["y0"])[:, 0] - g["solve"]).max() < 1e-10


def test_against_extended_precision():
    case = synthetic(1, 160, 2, 3, "accuracy", seed=3)
    co = coeffs_of(case, 0)
    t, diag, y = case["t"][0], case["diag"][0], case["y"][0]
    K = dense.dense_matrix(0.0, *co, *NO_GENERAL, t, diag)
    ld_mp, q_mp = dense.mp_logdet_quad(K, y)
    s = ref.RefSolver()
    s.compute(0.0, *co, *NO_GENERAL, t, diag)
    assert abs(s.log_determinant() - ld_mp) < 1e-12 * abs(ld_mp)
    assert abs(s.dot_solve(y) - q_mp) < 1e-12 * abs(q_mp)


def test_binary128_recurrence_is_pinned():
    """oracle/celerite_ref_quad.c -- the reference's recurrences carried in binary128, "the truth" the -m gpu tests
    attribute deviations with -- against (i) the mpmath 40-digit dense LDL^T (log det, quadratic form: it must be at
    least as close as the double-precision restatement, and within 1e-14), (ii) the double restatement on a long series
    (W, D, solve: agreement to the latter's rounding, a few 1e-13) and (iii) its failure contract (cholesky.h:176)."""
    case = synthetic(1, 160, 2, 3, "accuracy", seed=3)
    co = coeffs_of(case, 0)
    t, diag, y = case["t"][0], case["diag"][0], case["y"][0]
    K = dense.dense_matrix(0.0, *co, *NO_GENERAL, t, diag)
    ld_mp, q_mp = dense.mp_logdet_quad(K, y)
    W, D, x, ld, q = ref.quad_factor_solve(0.0, *co, t, diag, y)
    s = ref.RefSolver()
    s.compute(0.0, *co, *NO_GENERAL, t, diag)
    assert abs(ld - ld_mp) <= 1e-14 * abs(ld_mp) and abs(q - q_mp) <= 1e-14 * abs(q_mp)
    assert abs(ld - ld_mp) <= abs(s.log_determinant() - ld_mp) + 1e-15 * abs(ld_mp)
    assert np.allclose(x, np.linalg.solve(K, y), rtol=0, atol=1e-11 * np.max(np.abs(x)))
    # (the accuracy family's times reach 1.6e4: the double-precision product d t in the argument of cos / sin is itself
    #  rounded at 1e-12 absolute, which binary128 does not share -- the looser bar there)
    for JR, JC, family, tol in ((2, 3, "bench", 2e-12), (0, 2, "accuracy", 1e-10), (3, 0, "bench", 2e-12)):
        case = synthetic(1, 20000, JR, JC, family, seed=11 + JR)
        co = coeffs_of(case, 0)
        t, diag, y = case["t"][0], case["diag"][0], case["y"][0]
        W, D, x, ld, q = ref.quad_factor_solve(0.05, *co, t, diag, y)
        s = ref.RefSolver()
        s.compute(0.05, *co, *NO_GENERAL, t, diag)
        _, N, J, logdet, _, _, rW, rD = s.state()
        assert W.shape == rW.shape == (J, N)
        assert np.max(np.abs(W - rW)) <= tol * np.max(np.abs(rW)) and np.max(np.abs(D - rD) / np.abs(rD)) <= tol
        assert abs(ld - logdet) <= 1e-12 * abs(logdet) and abs(q - s.dot_solve(y)) <= 1e-12 * abs(q)
        xs = s.solve(y)[:, 0]
        assert np.max(np.abs(x - xs)) <= 10 * tol * np.max(np.abs(xs))
    bad = synthetic(1, 300, 1, 1, "bench", seed=5)
    co = list(coeffs_of(bad, 0))
    co[0] = -5.0 * np.abs(co[0])
    with pytest.raises(ref.RefLinAlgError):
        ref.quad_factor_solve(0.0, *co, bad["t"][0], np.zeros(300), bad["y"][0])
    with pytest.raises(ref.RefLinAlgError):
        s = ref.RefSolver()
        s.compute(0.0, *co, *NO_GENERAL, bad["t"][0], np.zeros(300))


def test_error_codes_and_edge_shapes():
    s = ref.RefSolver()
    e = np.empty(0)
    t = np.linspace(0, 1, 10)
    with pytest.raises(RuntimeError, match="dimension mismatch"):
        s.compute(0.0, np.ones(1), np.ones(2), e, e, e, e, *NO_GENERAL, t, np.ones(10))
    with pytest.raises(RuntimeError, match="dimension mismatch"):
        s.compute(0.0, np.ones(1), np.ones(1), e, e, e, e, *NO_GENERAL, t, np.ones(9))
    assert not s.computed()
    # not positive definite: negative amplitude, zero diagonal (tests/test_celerite.py:324-331)
    with pytest.raises(ref.RefLinAlgError):
        s.compute(0.0, np.array([-1.0]), np.array([0.1]), e, e, e, e, *NO_GENERAL, t, np.zeros(10))
    assert not s.computed()
    # J == 0 (jitter only): cholesky.h:90-95
    s.compute(0.5, e, e, e, e, e, e, *NO_GENERAL, t, np.ones(10))
    assert abs(s.log_determinant() - 10 * np.log(1.5)) < 1e-14
    assert abs(s.dot_solve(np.ones(10)) - 10 / 1.5) < 1e-14
    # N == 1
    s.compute(0.0, np.ones(1), np.ones(1), e, e, e, e, *NO_GENERAL, t[:1], np.ones(1))
    assert abs(s.log_determinant() - np.log(2.0)) < 1e-15
    with pytest.raises(RuntimeError, match="dimension mismatch"):
        s.dot_solve(np.ones(3))


def test_wide_kernel_and_zero_dt():
    """Width 26 / 30 (dynamic-width path) and a duplicated time stamp
    (tests/test_celerite.py:346-370, :393-404)."""
    golden = json.load(open(os.path.join(GOLDEN, "terms_golden.json")))
    co = [np.array(b) for b in golden["test_log_likelihood full kernel (width 26)"]["coefficients"]]
    np.random.seed(42)
    x = np.sort(np.random.rand(10))
    yerr = np.random.uniform(0.1, 0.5, len(x))
    y = np.sin(x)
    for gen in (NO_GENERAL, general_terms(x, np.random.rand)):
        s = ref.RefSolver()
        s.compute(0.0, *co, *gen, x, yerr ** 2)
        K = dense.dense_matrix(0.0, *co, *gen, x, yerr ** 2)
        ll0, ld0, q0 = dense.dense_log_likelihood(K, y)
        assert abs(s.log_determinant() - ld0) < 1e-9 and abs(s.dot_solve(y) - q0) < 1e-8
    ind = len(x) // 2
    x2 = np.concatenate((x[:ind], [x[ind]], x[ind:]))
    y2 = np.concatenate((y[:ind], [y[ind]], y[ind:]))
    e2 = np.concatenate((yerr[:ind], [yerr[ind]], yerr[ind:]))
    s = ref.RefSolver()
    s.compute(0.0, *co, *NO_GENERAL, x2, e2 ** 2)
    K = dense.dense_matrix(0.0, *co, *NO_GENERAL, x2, e2 ** 2)
    ll0, ld0, q0 = dense.dense_log_likelihood(K, y2)
    assert abs(s.log_determinant() - ld0) < 1e-9 and abs(s.dot_solve(y2) - q0) < 1e-8


def test_nyquist_singularity():
    """tests/test_celerite.py:501-525."""
    np.random.seed(4220)
    a, c, d = np.exp(1.0), 1e-6, 1.0  # ComplexTerm(1.0, log(1e-6), log(1.0))
    ts = np.array([0.0, 0.5, 1.0, 1.5])
    ts[1] += 1e-9 * np.random.randn()
    ts[2] += 1e-8 * np.random.randn()
    ts[3] += 1e-7 * np.random.randn()
    yerr = np.random.uniform(low=0.1, high=0.2, size=len(ts))
    y = np.random.randn(len(ts))
    e = np.empty(0)
    co = (e, e, np.array([a]), np.array([0.0]), np.array([c]), np.array([d]))
    ll, ld, q, st = ref.batch_log_likelihood(0.0, *[np.atleast_2d(v) if v.size else np.empty((1, 0)) for v in co],
                                             ts, yerr ** 2, y)
    K = dense.dense_matrix(0.0, *co, *NO_GENERAL, ts, yerr ** 2)
    assert abs(ll[0] - dense.dense_log_likelihood(K, y)[0]) < 1e-10


def test_predict_vs_dense():
    """tests/test_celerite.py:468-496 (mean only; the covariance is dense algebra)."""
    np.random.seed(42)
    x = np.linspace(1, 59, 300)
    t = np.sort(np.random.uniform(10, 50, 100))
    yerr = np.random.uniform(0.1, 0.5, len(t))
    y = np.sin(t)
    golden = json.load(open(os.path.join(GOLDEN, "terms_golden.json")))
    e = np.empty(0)
    co = (np.exp([0.1]), np.exp([0.5]), np.array([np.exp(0.6), np.exp(0.1)]),
          np.array([0.0, np.exp(0.05)]), np.array([np.exp(0.7), np.exp(0.5)]),
          np.array([np.exp(1.0), np.exp(-0.1)]))
    s = ref.RefSolver()
    s.compute(0.0, *co, *NO_GENERAL, t, yerr ** 2)
    K = dense.dense_matrix(0.0, *co, *NO_GENERAL, t, yerr ** 2)
    Ks = dense.kernel_value(*co, x[:, None] - t[None, :])
    assert np.abs(s.predict(y, x) - Ks @ np.linalg.solve(K, y)).max() < 1e-9


def test_batch_threads_agree():
    case = synthetic(6, 400, 2, 1, "bench", seed=5)
    one = ref.batch_log_likelihood(0.0, *coeffs_of(case), case["t"], case["diag"], case["y"], nthreads=1)
    four = ref.batch_log_likelihood(0.0, *coeffs_of(case), case["t"], case["diag"], case["y"], nthreads=4)
    for a, b in zip(one, four):
        assert np.array_equal(a, b)


# ---- gradient oracle (oracle/grad.py: solver.cpp:347-463 on dual numbers) ----------------
@pytest.mark.parametrize("with_general", [False, True])
@pytest.mark.parametrize("jitter", [0.0, 0.05])
def test_grad_oracle_value_and_finite_differences(with_general, jitter):
    """Pins the dual-number restatement the way the reference's own gradient test does
    (tests/test_celerite.py:452-481: finite differences, eps = 1.34e-7): its value
    equals the pinned log-likelihood (up to the reference's pi*log(N) constant) and
    every partial equals a central difference of it."""
    from oracle import grad as ograd

    np.random.seed(42)
    x = np.sort(np.random.rand(100))          # tests/test_celerite.py:427-430
    yerr = np.random.uniform(0.1, 0.5, len(x))
    y = np.sin(x)
    gen = general_terms(x, np.random.rand) if with_general else NO_GENERAL
    co = [np.array([1.5, 0.3]), np.array([0.7, 2.0]), np.array([1.0, 0.4]), np.array([0.1, 0.3]),
          np.array([1.2, 0.5]), np.array([3.0, 1.5])]

    def ll(jit, c):
        s = ref.RefSolver()
        s.compute(jit, *c, *gen, x, yerr ** 2)
        return -0.5 * (s.dot_solve(y) + s.log_determinant() + np.pi * np.log(len(x)))

    value, g = ograd.grad_log_likelihood(jitter, *co, *gen, x, y, yerr ** 2)
    assert g.shape == (13,)
    assert abs(value - ll(jitter, co)) <= 1e-12 * abs(value)
    eps = 1e-6
    fd = []
    for i in range(6):
        for j in range(2):
            cp = [c.copy() for c in co]
            cm = [c.copy() for c in co]
            cp[i][j] += eps
            cm[i][j] -= eps
            fd.append((ll(jitter, cp) - ll(jitter, cm)) / (2 * eps))
    assert np.allclose(g[1:], fd, rtol=2e-6, atol=1e-7)
    if jitter > 0:
        assert np.isclose(g[0], (ll(jitter + eps, co) - ll(jitter - eps, co)) / (2 * eps), rtol=2e-6)
    else:
        assert g[0] == 0.0                     # solver.cpp:379-389,419-426


def test_grad_oracle_raises_like_the_reference():
    from oracle import grad as ograd

    x = np.linspace(0, 1, 20)
    with pytest.raises(ograd.LinAlgError):
        ograd.grad_log_likelihood(0.0, [-3.0], [0.5], [], [], [], [], *NO_GENERAL, x, np.sin(x), np.zeros(20))


@pytest.mark.parametrize("log_sigma,ar,ma", [(-0.5, [0.1, 0.05, 0.01], [0.2, 0.1]), (0.3, [0.5, -0.2], [0.1]),
                                             (0.0, [1.0, 0.3, -0.4, 0.2, 0.05], [0.3, -0.1, 0.2]), (-1.0, [0.4], [])])
def test_carma_oracle_pinned_by_the_reference_test_identity_and_dense(log_sigma, ar, ma):
    """oracle/carma.py (restating carma.h) is pinned by what the reference's own test asserts
    (tests/test_celerite.py:22-42, first parameter set and its seeded inputs): the Kalman-filter likelihood
    equals the celerite likelihood of get_celerite_coeffs() -- the latter computed by the (pinned) Cholesky
    oracle -- and by a dense multivariate normal built from covariance(tau) (carma.h:255-272), which involves
    neither the filter nor the coefficient conversion."""
    from oracle import carma
    np.random.seed(42)
    t = np.sort(np.random.uniform(0, 5, 100))
    yerr = 0.1 + np.zeros_like(t)
    y = np.sin(t) + yerr * np.random.randn(len(t))
    cs = carma.CARMASolver(log_sigma, ar, ma)
    ll = cs.log_likelihood(t, y, yerr)
    params = cs.get_celerite_coeffs()
    r = ref.RefSolver()
    r.compute(0.0, *params, *NO_GENERAL, t, yerr ** 2)
    ll_celerite = -0.5 * (r.dot_solve(y) + r.log_determinant() + len(t) * np.log(2 * np.pi))
    assert abs(ll - ll_celerite) <= 1e-10 * abs(ll_celerite)          # (the reference asserts np.allclose)
    tau = np.abs(t[:, None] - t[None, :])
    K = np.vectorize(cs.covariance)(tau) + np.diag(yerr ** 2)
    _, ld = np.linalg.slogdet(K)
    ll_dense = -0.5 * (y.dot(np.linalg.solve(K, y)) + ld + len(t) * np.log(2 * np.pi))
    assert abs(ll - ll_dense) <= 1e-9 * abs(ll_dense)
    kv = dense.kernel_value(*params, tau[0]) if hasattr(dense, "kernel_value") else None
    if kv is not None:
        assert np.allclose(kv, np.vectorize(cs.covariance)(tau[0]), rtol=1e-9, atol=1e-12)


def test_carma_oracle_errors():
    from oracle import carma
    with pytest.raises(RuntimeError, match="dimension mismatch"):      # carma.h:59
        carma.CARMASolver(-0.5, [0.1], [0.2])
    cs = carma.CARMASolver(-0.5, [0.1, 0.05], [0.2])
    with pytest.raises(RuntimeError, match="dimension mismatch"):      # carma.h:223
        cs.log_likelihood(np.zeros(3), np.zeros(2), np.ones(3))
