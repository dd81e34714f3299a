# -*- coding: utf-8 -*-
"""`-m gpu`: the single-problem solver object and the GP front-end on an MI355X.

Two kinds of assertion:
  * the reference's own tests (tests/test_celerite.py), re-authored against
    `celerite_amd` -- same seeded inputs, same dense-LAPACK comparators, same
    exception / flag protocol;
  * parity with the CPU oracle on identical inputs at the north-star bar:
    |delta| / |value| <= 1e-10 on log_determinant and dot_solve (stated per
    assertion; observed ~1e-13).
Everything goes through the pybind11 module, i.e. through the C ABI."""
import json
import os
import pickle

import numpy as np
import pytest

import celerite_amd
from celerite_amd import GP, batch, terms
from celerite_amd.solver import get_kernel_value, LinAlgError
from oracle import dense, ref
from _cases import (COEFFS_W4, COEFFS_W10, COEFFS_DOT, COEFFS_PICKLE, COEFFS_CC_REAL, COEFFS_CC_COMP,
                    NO_GENERAL, FIRST_TUTORIAL_LOGLIKE, first_tutorial_case, general_terms,
                    logdet_case, solve_case, synthetic, coeffs_of, within)

pytestmark = pytest.mark.gpu
REL = 1e-10  # north-star tolerance on log_determinant and dot_solve
E0 = np.empty(0)


def rel(a, b):
    return abs(a - b) / abs(b)


# ---- the reference's tests, re-authored ------------------------------------
@pytest.mark.parametrize("coeffs", [COEFFS_W4, COEFFS_W10])
def test_log_determinant(coeffs):  # tests/test_celerite.py:45-85
    t, diag = logdet_case()
    s = celerite_amd.CholeskySolver()
    s.compute(0.0, *coeffs, *NO_GENERAL, t, diag)
    K = get_kernel_value(*coeffs, t[:, None] - t[None, :])
    K[np.diag_indices_from(K)] += diag
    assert np.allclose(s.log_determinant(), np.linalg.slogdet(K)[1])


@pytest.mark.parametrize("with_general", [True, False])
@pytest.mark.parametrize("coeffs", [COEFFS_W4, COEFFS_W10])
def test_solve(coeffs, with_general):  # tests/test_celerite.py:88-151
    t, diag, b, (A, U, V) = solve_case(with_general)
    s = celerite_amd.CholeskySolver()
    with pytest.raises(RuntimeError):
        s.log_determinant()
    with pytest.raises(RuntimeError):
        s.dot_solve(b)
    s.compute(0.0, *coeffs, A, U, V, t, diag)
    K = get_kernel_value(*coeffs, t[:, None] - t[None, :])
    K[np.diag_indices_from(K)] += diag
    if len(A):
        K[np.diag_indices_from(K)] += A
        K += np.tril(np.dot(U.T, V), -1) + np.triu(np.dot(V.T, U), 1)
    x = s.solve(b)
    assert x.shape == (len(t), 1)  # always 2-D, like the reference
    assert np.allclose(x.T, np.linalg.solve(K, b))
    b5 = np.random.randn(len(t), 5)
    assert np.allclose(s.solve(b5), np.linalg.solve(K, b5))
    # parity with the oracle at the north-star bar
    r = ref.RefSolver()
    r.compute(0.0, *coeffs, A, U, V, t, diag)
    assert rel(s.log_determinant(), r.log_determinant()) <= REL
    assert rel(s.dot_solve(b), r.dot_solve(b)) <= REL
    assert np.abs(s.solve(b5) - r.solve(b5)).max() <= 1e-10 * np.abs(r.solve(b5)).max()


@pytest.mark.parametrize("with_general", [True, False])
def test_dot(with_general):  # tests/test_celerite.py:154-192
    s = celerite_amd.CholeskySolver()
    np.random.seed(42)
    t = np.sort(np.random.rand(500))
    b = np.random.randn(len(t), 5)
    K = get_kernel_value(*COEFFS_DOT, t[:, None] - t[None, :])
    if with_general:
        A, U, V = general_terms(t, np.random.rand)
        K[np.diag_indices_from(K)] += A
        K += np.tril(np.dot(U.T, V), -1) + np.triu(np.dot(V.T, U), 1)
    else:
        A, U, V = NO_GENERAL
    x = s.dot(0.0, *COEFFS_DOT, A, U, V, t, b)
    assert np.allclose(np.dot(K, b), x)
    x1 = s.dot(0.0, *COEFFS_DOT, A, U, V, t, b[:, 0])
    assert x1.shape == (len(t), 1) and np.allclose(x1[:, 0], x[:, 0])


@pytest.mark.parametrize("with_general", [True, False])
def test_dot_L(with_general):  # tests/test_celerite.py:194-235
    s = celerite_amd.CholeskySolver()
    np.random.seed(42)
    t = np.sort(np.random.rand(5))
    b = np.random.randn(len(t), 5)
    yerr = np.random.uniform(0.1, 0.5, len(t))
    K = get_kernel_value(*COEFFS_DOT, t[:, None] - t[None, :])
    K[np.diag_indices_from(K)] += yerr ** 2
    if with_general:
        A, U, V = general_terms(t, np.random.rand)
        K[np.diag_indices_from(K)] += A
        K += np.tril(np.dot(U.T, V), -1) + np.triu(np.dot(V.T, U), 1)
    else:
        A, U, V = NO_GENERAL
    s.compute(0.0, *COEFFS_DOT, A, U, V, t, yerr ** 2)
    assert np.allclose(np.dot(np.linalg.cholesky(K), b), s.dot_L(b))


@pytest.mark.parametrize("with_general", [True, False])
def test_pickle(with_general):  # tests/test_celerite.py:237-289
    s = celerite_amd.CholeskySolver()
    np.random.seed(42)
    t = np.sort(np.random.rand(500))
    diag = np.random.uniform(0.1, 0.5, len(t))
    y = np.sin(t)
    A, U, V = general_terms(t, np.random.rand) if with_general else NO_GENERAL

    def compare(s1, s2):
        assert s1.computed() == s2.computed()
        if not s1.computed():
            return
        assert np.allclose(s1.log_determinant(), s2.log_determinant())
        assert np.allclose(s1.dot_solve(y), s2.dot_solve(y))

    compare(s, pickle.loads(pickle.dumps(s, -1)))
    s.compute(0.0, *COEFFS_PICKLE, A, U, V, t, diag)
    s2 = pickle.loads(pickle.dumps(s, -1))
    compare(s, s2)
    # the pickled state is the reference's 8-tuple (solver.cpp:36-42), bit-identical
    st, st2 = s.__getstate__(), s2.__getstate__()
    J = 4 + (4 if with_general else 0)
    assert st[:3] == (True, 500, J) and st[4].shape == (J, 499) and st[6].shape == (J, 500)
    for a, b in zip(st[4:], st2[4:]):
        assert np.array_equal(a, b)
    # ... and equals the oracle's factor
    r = ref.RefSolver()
    r.compute(0.0, *COEFFS_PICKLE, A, U, V, t, diag)
    rs = r.state()
    assert np.allclose(st[4], rs[4], rtol=1e-13, atol=0)       # phi
    assert np.allclose(st[5], rs[5], rtol=1e-12, atol=1e-15)   # u
    assert np.allclose(st[6], rs[6], rtol=1e-9, atol=1e-12)    # W
    assert np.allclose(st[7], rs[7], rtol=1e-11, atol=0)       # D

    kernel = terms.RealTerm(0.5, 0.1)
    kernel += terms.ComplexTerm(0.6, 0.7, 1.0)
    gp1 = GP(kernel)
    gp1.compute(t, diag)
    gp2 = pickle.loads(pickle.dumps(gp1, -1))
    assert np.allclose(gp1.log_likelihood(y), gp2.log_likelihood(y))


@pytest.mark.parametrize("with_general", [True, False])
def test_log_likelihood(with_general):  # tests/test_celerite.py:311-404
    np.random.seed(42)
    x = np.sort(np.random.rand(10))
    yerr = np.random.uniform(0.1, 0.5, len(x))
    y = np.sin(x)
    A, U, V = general_terms(x, np.random.rand) if with_general else NO_GENERAL

    class NPDTerm(terms.Term):
        parameter_names = ("par1", )

        def get_real_coefficients(self, params):
            return [params[0]], [0.1]

    gp = GP(NPDTerm(-1.0))
    with pytest.raises(LinAlgError):
        gp.compute(x, 0.0)
    with pytest.raises(LinAlgError):
        gp.log_likelihood(y)
    assert np.isinf(gp.log_likelihood(y, quiet=True))

    kernel = terms.RealTerm(0.1, 0.5)
    gp = GP(kernel)
    with pytest.raises(RuntimeError):
        gp.log_likelihood(y)

    termlist = [(0.1 + 10. / j, 0.5 + 10. / j) for j in range(1, 4)]
    termlist += [(1.0 + 10. / j, 0.01 + 10. / j, 0.5, 0.01) for j in range(1, 10)]
    termlist += [(0.6, 0.7, 1.0), (0.3, 0.05, 0.5, 0.6)]
    for term in termlist:  # widths 2 .. 26 (30 with general terms): every kernel path
        kernel += terms.ComplexTerm(*term) if len(term) > 2 else terms.RealTerm(*term)
        gp = GP(kernel)
        assert gp.computed is False
        with pytest.raises(ValueError):
            gp.compute(np.random.rand(len(x)), yerr)
        gp.compute(x, yerr, A=A, U=U, V=V)
        assert gp.computed is True
        assert gp.dirty is False
        ll = gp.log_likelihood(y)
        K = gp.get_matrix(include_diagonal=True)
        ll0 = -0.5 * np.dot(y, np.linalg.solve(K, y))
        ll0 -= 0.5 * np.linalg.slogdet(K)[1]
        ll0 -= 0.5 * len(x) * np.log(2 * np.pi)
        assert np.allclose(ll, ll0)

    gp.set_parameter_vector(gp.get_parameter_vector())
    assert gp.dirty is True
    assert gp.computed is False

    gp.compute(x, yerr, A=A, U=U, V=V)
    ll1 = gp.log_likelihood(y)
    params = gp.get_parameter_vector()
    params[0] += 10.0
    gp.set_parameter_vector(params)
    gp.compute(x, yerr, A=A, U=U, V=V)
    ll2 = gp.log_likelihood(y)
    assert not np.allclose(ll1, ll2)

    gp[1] += 10.0
    assert gp.dirty is True
    gp.compute(x, yerr, A=A, U=U, V=V)
    ll3 = gp.log_likelihood(y)
    assert not np.allclose(ll2, ll3)

    ind = len(x) // 2  # zero delta t
    x = np.concatenate((x[:ind], [x[ind]], x[ind:]))
    y = np.concatenate((y[:ind], [y[ind]], y[ind:]))
    yerr = np.concatenate((yerr[:ind], [yerr[ind]], yerr[ind:]))
    gp.compute(x, yerr)
    ll = gp.log_likelihood(y)
    K = gp.get_matrix(include_diagonal=True)
    ll0 = -0.5 * np.dot(y, np.linalg.solve(K, y))
    ll0 -= 0.5 * np.linalg.slogdet(K)[1]
    ll0 -= 0.5 * len(x) * np.log(2 * np.pi)
    assert np.allclose(ll, ll0)


GRAD_KERNELS = [  # tests/test_celerite.py:407-423
    lambda: terms.RealTerm(log_a=0.1, log_c=0.5),
    lambda: terms.RealTerm(log_a=0.1, log_c=0.5) + terms.RealTerm(log_a=-0.1, log_c=0.7),
    lambda: terms.ComplexTerm(log_a=0.1, log_c=0.5, log_d=0.1),
    lambda: terms.ComplexTerm(log_a=0.1, log_b=-0.2, log_c=0.5, log_d=0.1),
    lambda: terms.JitterTerm(log_sigma=0.1),
    lambda: terms.SHOTerm(log_S0=0.1, log_Q=-1, log_omega0=0.5) + terms.JitterTerm(log_sigma=0.1),
    lambda: terms.SHOTerm(log_S0=0.1, log_Q=-1, log_omega0=0.5),
    lambda: terms.SHOTerm(log_S0=0.1, log_Q=1.0, log_omega0=0.5),
    lambda: terms.SHOTerm(log_S0=0.1, log_Q=1.0, log_omega0=0.5) + terms.RealTerm(log_a=0.1, log_c=0.4),
    lambda: terms.SHOTerm(log_S0=0.1, log_Q=1.0, log_omega0=0.5) * terms.RealTerm(log_a=0.1, log_c=0.4),
]


@pytest.mark.parametrize("make_kernel", GRAD_KERNELS)
@pytest.mark.parametrize("with_general", [False, True])
def test_grad_log_likelihood(make_kernel, with_general):  # tests/test_celerite.py:425-446
    """The solver-level gradient (value + d/d(jitter, coefficients), solver.cpp:347-463)
    against the dual-number oracle on the reference test's inputs; then the GP-level gradient
    against central differences of the GP's own log-likelihood (:448-465)."""
    from oracle import grad as ograd

    kernel = make_kernel()
    np.random.seed(42)
    x = np.sort(np.random.rand(100))
    yerr = np.random.uniform(0.1, 0.5, len(x))
    y = np.sin(x)
    if with_general:
        A, U, V = general_terms(x, np.random.rand)
    else:
        A, U, V = NO_GENERAL
    gp = GP(kernel)
    gp.compute(x, yerr, A=A, U=U, V=V)
    args = (kernel.jitter,) + tuple(kernel.coefficients) + (A, U, V, x, y, yerr ** 2)
    value, g = gp.solver.grad_log_likelihood(*args)
    v0, g0 = ograd.grad_log_likelihood(*args)
    assert g.shape == g0.shape == (1 + 2 * len(args[1]) + 4 * len(args[3]),)
    assert abs(value - v0) <= 1e-11 * abs(v0)
    assert np.max(np.abs(g - g0)) <= 1e-9 * max(1.0, np.max(np.abs(g0)))
    # the value is the log-likelihood up to the reference's odd constant (solver.cpp:415)
    ll = gp.log_likelihood(y)
    assert np.isclose(value + 0.5 * np.pi * np.log(len(x)), ll + 0.5 * len(x) * np.log(2 * np.pi), rtol=1e-12)
    # the GP-level gradient (celerite.py:221-305): the reference needs autograd for the chain rule to the kernel's
    # log-parameters; here the built-in terms carry their own Jacobians (terms._dual_coefficients), so the branch the
    # reference only runs WITH autograd (tests/test_celerite.py:448-465: central differences of log_likelihood, with and
    # without a fitted mean) runs always
    eps = 1.34e-7
    for fit_mean in (True, False):
        gp = GP(kernel, fit_mean=fit_mean)
        gp.compute(x, yerr, A=A, U=U, V=V)
        _, grad = gp.grad_log_likelihood(y)
        grad0 = np.empty_like(grad)
        v = gp.get_parameter_vector()
        for i, pval in enumerate(v):
            v[i] = pval + eps
            gp.set_parameter_vector(v)
            ll = gp.log_likelihood(y)
            v[i] = pval - eps
            gp.set_parameter_vector(v)
            ll -= gp.log_likelihood(y)
            grad0[i] = 0.5 * ll / eps
            v[i] = pval
        gp.set_parameter_vector(v)
        assert np.allclose(grad, grad0), (fit_mean, grad, grad0)


def test_grad_log_likelihood_wide_and_long():
    """Widths 17 and 40 (both register layouts above 16) at N = 700, zero-jitter rule,
    dimension checks and the exception of an indefinite matrix."""
    from oracle import grad as ograd

    rng = np.random.RandomState(3)
    N = 700
    x = np.sort(rng.uniform(0, 30, N))
    diag = rng.uniform(0.1, 0.3, N)
    y = rng.randn(N)
    s = celerite_amd.CholeskySolver()
    for JR, JC in ((1, 8), (4, 18)):
        co = (np.exp(rng.uniform(-1, 1, JR)), np.exp(rng.uniform(-2, 0, JR)), np.exp(rng.uniform(-1, 0, JC)),
              0.2 * rng.rand(JC), np.exp(rng.uniform(-2, 0, JC)), np.exp(rng.uniform(-1, 1.5, JC)))
        for jitter in (0.0, 0.3):
            args = (jitter,) + co + NO_GENERAL + (x, y, diag)
            value, g = s.grad_log_likelihood(*args)
            v0, g0 = ograd.grad_log_likelihood(*args)
            assert abs(value - v0) <= 1e-11 * abs(v0)
            assert np.max(np.abs(g - g0) / (1.0 + np.abs(g0))) <= 1e-9
            assert (g[0] == 0.0) == (jitter == 0.0)   # solver.cpp:379-389,419-426
    co = (np.array([-3.0]), np.array([0.5]), np.empty(0), np.empty(0), np.empty(0), np.empty(0))
    with pytest.raises(celerite_amd.solver.LinAlgError):
        s.grad_log_likelihood(0.0, *co, *NO_GENERAL, x, y, np.zeros(N))
    with pytest.raises(RuntimeError):
        s.grad_log_likelihood(0.0, *co, *NO_GENERAL, x, y[:-1], diag)


def test_gp_gradient_with_many_terms():
    """``GP.grad_log_likelihood`` (celerite.py:221-305) for a kernel of 34 SHO terms + jitter -- width 68, above the
    wave-per-partial kernel's 64 -- through the workgroup-per-partial kernel: the reference test's protocol
    (tests/test_celerite.py:448-465: central differences of the GP's own log-likelihood over every kernel parameter)."""
    rng = np.random.RandomState(4)
    kernel = terms.JitterTerm(log_sigma=-1.0)
    for k in range(34):
        kernel += terms.SHOTerm(log_S0=rng.uniform(-2.0, 0.0), log_Q=rng.uniform(0.5, 2.0), log_omega0=rng.uniform(-1.0, 2.5))
    x = np.sort(rng.uniform(0, 20, 160))
    yerr = rng.uniform(0.1, 0.3, len(x))
    y = np.sin(x) + 0.1 * rng.randn(len(x))
    gp = GP(kernel)
    gp.compute(x, yerr)
    assert sum(len(c) for c in kernel.coefficients[:2]) // 2 + 2 * len(kernel.coefficients[2]) == 68
    value, grad = gp.grad_log_likelihood(y)
    # (the value carries the reference's own constant, pi log N instead of N log 2 pi: solver.cpp:415)
    within("GP gradient with 34 SHO terms: value vs log_likelihood",
           abs((value + 0.5 * np.pi * np.log(len(x))) - (gp.log_likelihood(y) + 0.5 * len(x) * np.log(2 * np.pi))) / abs(value), 1e-11)
    v = gp.get_parameter_vector()
    assert grad.shape == v.shape == (1 + 3 * 34,)
    eps = 1.34e-6
    for i in list(range(0, len(v), 9)) + [len(v) - 1]:
        pval = v[i]
        v[i] = pval + eps
        gp.set_parameter_vector(v)
        ll = gp.log_likelihood(y)
        v[i] = pval - eps
        gp.set_parameter_vector(v)
        ll -= gp.log_likelihood(y)
        v[i] = pval
        gp.set_parameter_vector(v)
        within("GP gradient with 34 SHO terms: vs central differences (of 1 + |g|)", abs(grad[i] - 0.5 * ll / eps) / (1.0 + abs(grad[i])), 1e-5, i)


def _grad_case(rng, JR, JC, JG, N):
    x = np.sort(rng.uniform(0, 30, N))
    diag = rng.uniform(0.1, 0.3, N)
    y = rng.randn(N)
    co = (np.exp(rng.uniform(-1, 1, JR)), np.exp(rng.uniform(-2, 0, JR)), np.exp(rng.uniform(-1, 0, JC)),
          0.2 * rng.rand(JC), np.exp(rng.uniform(-2, 0, JC)), np.exp(rng.uniform(-1, 1.5, JC)))
    if JG:
        U = np.vander(x - np.mean(x), JG).T * 1e-3
        V = U * rng.rand(JG)[:, None]
        gen = (np.sum(U * V, axis=0) + 1e-8, U, V)
    else:
        gen = NO_GENERAL
    return co, gen, x, y, diag


def test_grad_log_likelihood_above_width_64():
    """``grad_log_likelihood`` at any width (the reference's dual-number solver takes any J: solver.cpp:347-463,
    cholesky.h:203): above 64 one workgroup per partial with S and its tangent in an HBM / L2 workspace
    (csrc/grad_any_kernels.hip).  Against the dual-number oracle at widths 70 (2 real + 34 complex: 141 partials) and 67
    (general terms as constant rows); forced at width 40 against the oracle AND the wave-per-partial kernel it extends;
    at width 130 and 260 (rounds of 256 partials) against central differences of the device's own log-likelihood;
    the zero-jitter rule, the indefinite matrix and the dimension checks as below width 64."""
    from oracle import grad as ograd

    rng = np.random.RandomState(11)
    s = celerite_amd.CholeskySolver()
    for JR, JC, JG, N in ((2, 34, 0, 120), (1, 30, 6, 100)):
        co, gen, x, y, diag = _grad_case(rng, JR, JC, JG, N)
        for jitter in (0.0, 0.3):
            args = (jitter,) + co + gen + (x, y, diag)
            value, g = s.grad_log_likelihood(*args)
            v0, g0 = ograd.grad_log_likelihood(*args)
            within("gradient above width 64 (width %d): value vs the dual-number oracle" % (JR + 2 * JC + JG), abs(value - v0) / abs(v0), 1e-11)
            within("gradient above width 64 (width %d): partials vs the dual-number oracle, of 1 + |g|" % (JR + 2 * JC + JG),
                   np.max(np.abs(g - g0) / (1.0 + np.abs(g0))), 1e-9)
            assert (g[0] == 0.0) == (jitter == 0.0)   # solver.cpp:379-389,419-426
    # the same kernel forced at width 40, where the wave-per-partial kernel exists
    co, gen, x, y, diag = _grad_case(rng, 4, 18, 0, 700)
    args = (0.2,) + co + gen + (x, y, diag)
    with batch.option("CLR_GRAD_SEQUENTIAL"):
        v_wave, g_wave = s.grad_log_likelihood(*args)
        with batch.option("CLR_GRAD_ANY_WIDTH"):
            v_any, g_any = s.grad_log_likelihood(*args)
    v0, g0 = ograd.grad_log_likelihood(*args)
    within("any-width gradient kernel forced at width 40: value vs oracle", abs(v_any - v0) / abs(v0), 1e-11)
    within("any-width gradient kernel forced at width 40: partials vs oracle, of 1 + |g|", np.max(np.abs(g_any - g0) / (1.0 + np.abs(g0))), 1e-9)
    within("any-width gradient kernel forced at width 40: partials vs the wave-per-partial kernel, of 1 + |g|",
           np.max(np.abs(g_any - g_wave) / (1.0 + np.abs(g_wave))), 1e-10)
    # widths 130 / 260: central differences of the device's own log-likelihood in a few directions
    for JR, JC, N in ((2, 64, 300), (4, 128, 150)):
        co, gen, x, y, diag = _grad_case(rng, JR, JC, 0, N)
        jitter = 0.1
        value, g = s.grad_log_likelihood(jitter, *co, *gen, x, y, diag)
        assert g.shape == (1 + 2 * JR + 4 * JC,) and np.all(np.isfinite(g))

        def ll(jit, cc):
            s.compute(jit, *cc, *gen, x, diag)
            return -0.5 * (s.dot_solve(y) + s.log_determinant() + np.pi * np.log(N))   # solver.cpp:415

        within("gradient at width %d: value vs compute + dot_solve" % (JR + 2 * JC), abs(value - ll(jitter, co)) / abs(value), 1e-11)
        offs = np.cumsum([1, JR, JR, JC, JC, JC])          # a_real, c_real, a_comp, b_comp, c_comp, d_comp
        picks = [(-1, 0)] + [(k, j) for k in range(6) for j in ((0, len(co[k]) - 1) if len(co[k]) > 1 else (0,))]
        for k, j in picks:
            base = jitter if k < 0 else co[k][j]
            h = 1e-6 * max(abs(base), 1e-2)
            def shifted(d):
                if k < 0:
                    return ll(jitter + d, co)
                cc = [c.copy() for c in co]
                cc[k][j] += d
                return ll(jitter, cc)
            fd = (shifted(h) - shifted(-h)) / (2.0 * h)
            got = g[0] if k < 0 else g[offs[k] + j]
            within("gradient at width %d: partials vs central differences of the device log-likelihood, of 1 + |g|" % (JR + 2 * JC),
                   abs(got - fd) / (1.0 + abs(fd)), 2e-5, (k, j))
    co = (np.array([-300.0]), np.array([0.5]), np.exp(rng.uniform(-1, 0, 33)), np.zeros(33), np.exp(rng.uniform(-2, 0, 33)), np.exp(rng.uniform(-1, 1, 33)))
    x = np.sort(rng.uniform(0, 30, 200))
    with pytest.raises(celerite_amd.solver.LinAlgError):
        s.grad_log_likelihood(0.0, *co, *NO_GENERAL, x, rng.randn(200), np.zeros(200))
    with pytest.raises(RuntimeError):
        s.grad_log_likelihood(0.0, *co, *NO_GENERAL, x, rng.randn(199), np.full(200, 0.1))


@pytest.mark.parametrize("JR,JC,JG,N", [(4, 4, 0, 4200), (0, 8, 0, 20000), (2, 5, 0, 9000), (0, 16, 0, 8000), (6, 13, 0, 5000),
                                        (2, 3, 4, 6000), (0, 8, 3, 5000), (1, 0, 2, 4096),
                                        (2, 16, 0, 9000), (1, 24, 0, 6000), (0, 32, 0, 8000)])
def test_grad_log_likelihood_at_widths_9_to_32_and_with_general_terms_is_parallel_in_n(JR, JC, JG, N):
    """From N = 4096 on, widths 9..32 (round 6: to 64) -- and any celerite width with general terms up to a total width of 32 -- run
    the wide scan + chunk-wise forward-mode tangents on a one-problem plan (csrc/wide_grad_kernels.hip; the reference's AD
    handles any width and J_general, solver.cpp:347-463, general rows :393-399): same numbers as the sequential tangent
    kernel, the dual-number oracle on the shortest cases, the zero-jitter rule, LinAlgError for an indefinite matrix."""
    from oracle import grad as ograd
    import os
    from _cases import within

    rng = np.random.RandomState(7 + JR + 3 * JC + JG)
    x = np.sort(rng.uniform(0, 0.05 * N, N))
    diag = rng.uniform(0.1, 0.3, N)
    y = rng.randn(N)
    if JG:
        z = (x - x.mean()) / (x.max() - x.min())
        U = np.vander(z, JG).T.copy()
        V = U * rng.rand(JG)[:, None]
        gen = (np.sum(U * V, axis=0) + 1e-8, U, V)
    else:
        gen = NO_GENERAL
    s = celerite_amd.CholeskySolver()
    for trial in range(2):   # second call: new coefficients, the plan and its series are reused
        co = (np.exp(rng.uniform(-1, 1, JR)), np.exp(rng.uniform(-2, 0, JR)), np.exp(rng.uniform(-1, 0, JC)),
              0.2 * rng.rand(JC), np.exp(rng.uniform(-2, 0, JC)), np.exp(rng.uniform(-1, 1.5, JC)))
        args = (0.1 * trial,) + co + gen + (x, y, diag)
        value, g = s.grad_log_likelihood(*args)
        batch.set_option("CLR_GRAD_SEQUENTIAL", "1")
        try:
            v1, g1 = celerite_amd.CholeskySolver().grad_log_likelihood(*args)
        finally:
            batch.set_option("CLR_GRAD_SEQUENTIAL", None)
        within("wide / general gradient, object API: value vs sequential kernel", abs(value - v1) / abs(v1), 1e-12, (JR, JC, JG))
        within("wide / general gradient, object API: partials vs sequential kernel (of the largest)",
               np.max(np.abs(g - g1)) / np.max(np.abs(g1)), 1e-10, (JR, JC, JG))
        assert (g[0] == 0.0) == (trial == 0)
        if N <= 4200:
            v0, g0 = ograd.grad_log_likelihood(*args)
            within("wide / general gradient, object API: value vs oracle", abs(value - v0) / abs(v0), 1e-11)
            within("wide / general gradient, object API: partials vs oracle (of the largest)",
                   np.max(np.abs(g - g0)) / np.max(np.abs(g0)), 1e-10)
    bad = list(co)
    bad[0 if JR else 2] = -50.0 * bad[0 if JR else 2]
    with pytest.raises(celerite_amd.solver.LinAlgError):
        s.grad_log_likelihood(0.0, *bad, *gen, x, y, diag)


@pytest.mark.parametrize("JR,JC,N", [(2, 3, 3000), (1, 1, 20000), (0, 4, 6000), (3, 0, 2048)])
def test_grad_log_likelihood_of_a_long_series_is_parallel_in_n(JR, JC, N):
    """From N = 1024 on (widths 1..8, no general terms) CholeskySolver.grad_log_likelihood runs the scan + the
    chunk-wise tangents on a one-problem plan (csrc/clr_grad_core.h): same numbers as the sequential tangent kernel,
    the oracle on the shortest case, the series kept between calls, LinAlgError for an indefinite matrix."""
    from oracle import grad as ograd
    import os

    rng = np.random.RandomState(11 + JR)
    x = np.sort(rng.uniform(0, 0.6 * N, N))
    diag = rng.uniform(0.1, 0.3, N)
    y = rng.randn(N)
    s = celerite_amd.CholeskySolver()
    for trial in range(2):   # second call: new coefficients, the plan and its series are reused
        co = (np.exp(rng.uniform(-1, 1, JR)), np.exp(rng.uniform(-2, 0, JR)), np.exp(rng.uniform(-1, 0, JC)),
              0.2 * rng.rand(JC), np.exp(rng.uniform(-2, 0, JC)), np.exp(rng.uniform(-1, 1.5, JC)))
        args = (0.1 * trial,) + co + NO_GENERAL + (x, y, diag)
        value, g = s.grad_log_likelihood(*args)
        batch.set_option("CLR_GRAD_SEQUENTIAL", "1")
        try:
            v1, g1 = celerite_amd.CholeskySolver().grad_log_likelihood(*args)
        finally:
            batch.set_option("CLR_GRAD_SEQUENTIAL", None)
        assert abs(value - v1) <= 1e-11 * abs(v1)
        assert np.max(np.abs(g - g1)) <= 1e-8 * np.max(np.abs(g1)), np.max(np.abs(g - g1)) / np.max(np.abs(g1))
        assert (g[0] == 0.0) == (trial == 0)
        if N <= 3000:
            v0, g0 = ograd.grad_log_likelihood(*args)
            assert abs(value - v0) <= 1e-11 * abs(v0)
            assert np.max(np.abs(g - g0)) <= 1e-8 * np.max(np.abs(g0))
    bad = list(co)
    bad[0 if JR else 2] = -50.0 * bad[0 if JR else 2]
    with pytest.raises(celerite_amd.solver.LinAlgError):
        s.grad_log_likelihood(0.0, *bad, *NO_GENERAL, x, y, diag)


@pytest.mark.parametrize("N", [256, 700, 5000, 40000])
@pytest.mark.parametrize("shape", ["real", "w4", "w8", "w4+general"])
def test_long_series_sweeps_are_chunked_scans(N, shape):
    """dot_solve / solve on a stored factor switch to chunked scans over n at N >= 256 and
    width <= 8 (csrc/sweep_kernels.hip): same numbers as the oracle's sequential sweeps
    (cholesky.h:218-401), several right-hand sides, general terms included."""
    rng = np.random.RandomState(N % 97)
    t = np.sort(rng.uniform(0, N / 50.0, N))
    diag = rng.uniform(0.1, 0.4, N)
    gen = NO_GENERAL
    if shape == "real":
        co = (np.array([1.3]), np.array([0.5]), np.empty(0), np.empty(0), np.empty(0), np.empty(0))
    elif shape == "w8":
        co = (np.array([1.3, 0.4]), np.array([0.5, 0.05]), np.array([1.0, 0.3, 0.6]), np.array([0.1, 0.0, 0.2]),
              np.array([0.3, 0.8, 0.1]), np.array([1.0, 2.5, 0.4]))
    else:
        co = COEFFS_W4
        if shape == "w4+general":
            U = np.vander((t - t.mean()) / (t.max() - t.min()), 4).T
            V = U * rng.rand(4)[:, None]
            gen = (np.sum(U * V, axis=0) + 1e-8, U, V)
    s = celerite_amd.CholeskySolver()
    r = ref.RefSolver()
    s.compute(0.0, *co, *gen, t, diag)
    r.compute(0.0, *co, *gen, t, diag)
    b = rng.randn(N, 3)
    q, q0 = s.dot_solve(b[:, 0]), r.dot_solve(b[:, 0])
    assert abs(q - q0) <= 1e-12 * abs(q0)
    x, x0 = s.solve(b), r.solve(b)
    assert x.shape == (N, 3)
    assert np.max(np.abs(x - x0)) <= 1e-11 * np.max(np.abs(x0))
    x1 = s.solve(b[:, 1])
    assert np.max(np.abs(x1[:, 0] - x0[:, 1])) <= 1e-11 * np.max(np.abs(x0))
    yl, yl0 = s.dot_L(b), r.dot_L(b)                     # cholesky.h:409-431, diagonal scan
    assert np.max(np.abs(yl - yl0)) <= 1e-12 * np.max(np.abs(yl0))
    if shape != "w4+general":                            # predict: cholesky.h:599-698
        y = np.sin(t) + 0.1 * b[:, 0]
        for xs in (np.linspace(t[0] - 1.0, t[-1] + 1.0, 777),
                   np.sort(np.concatenate([t[::37], t[:5], rng.uniform(t[0], t[-1], 200)]))):
            p, p0 = s.predict(y, xs), r.predict(y, xs)
            assert np.max(np.abs(p - p0)) <= 1e-10 * max(1.0, np.max(np.abs(p0)))


def test_predict():  # tests/test_celerite.py:468-496
    np.random.seed(42)
    x = np.linspace(1, 59, 300)
    t = np.sort(np.random.uniform(10, 50, 100))
    yerr = np.random.uniform(0.1, 0.5, len(t))
    y = np.sin(t)
    kernel = terms.RealTerm(0.1, 0.5)
    for term in [(0.6, 0.7, 1.0), (0.1, 0.05, 0.5, -0.1)]:
        kernel += terms.ComplexTerm(*term)
    gp = GP(kernel)
    gp.compute(t, yerr)
    K = gp.get_matrix(include_diagonal=True)
    Ks = gp.get_matrix(x, t)
    true_mu = np.dot(Ks, np.linalg.solve(K, y))
    true_cov = gp.get_matrix(x, x) - np.dot(Ks, np.linalg.solve(K, Ks.T))
    mu, cov = gp.predict(y, x)
    _, var = gp.predict(y, x, return_var=True)
    assert np.allclose(mu, true_mu)
    assert np.allclose(cov, true_cov)
    assert np.allclose(var, np.diag(true_cov))
    mu0, cov0 = gp.predict(y, t)
    mu, cov = gp.predict(y)
    assert np.allclose(mu0, mu)
    assert np.allclose(cov0, cov)


def test_nyquist_singularity():  # tests/test_celerite.py:501-525
    np.random.seed(4220)
    gp = GP(terms.ComplexTerm(1.0, np.log(1e-6), np.log(1.0)))
    ts = np.array([0.0, 0.5, 1.0, 1.5])
    ts[1] = ts[1] + 1e-9 * np.random.randn()
    ts[2] = ts[2] + 1e-8 * np.random.randn()
    ts[3] = ts[3] + 1e-7 * np.random.randn()
    yerr = np.random.uniform(low=0.1, high=0.2, size=len(ts))
    y = np.random.randn(len(ts))
    gp.compute(ts, yerr)
    llgp = gp.log_likelihood(y)
    K = gp.get_matrix(ts)
    K[np.diag_indices_from(K)] += yerr ** 2.0
    ll = (-0.5 * np.dot(y, np.linalg.solve(K, y)) - 0.5 * np.linalg.slogdet(K)[1]
          - 0.5 * len(y) * np.log(2.0 * np.pi))
    assert np.allclose(ll, llgp)


# ---- BASELINE configs[0] as BASELINE words it ------------------------------------
def test_config0_gp_log_likelihood_against_the_oracle():
    """BASELINE configs[0]: one series of N = 1000 samples, 1 real + 1 SHO term (width 3), ``GP.log_likelihood`` on the
    tests/test_celerite.py path (:311-404: kernel -> GP -> compute -> log_likelihood).  The HIP product is driven
    through ``terms.RealTerm + terms.SHOTerm -> GP.compute -> GP.log_likelihood``; the comparator is the CPU oracle
    (``ref.RefSolver``: cholesky.h:41-210, :326-401) on the very arrays the GP hands to its solver, at 1e-10; the
    batched entry on a table of draws of the same kernel is held against the oracle as well (not against the object
    API)."""
    from celerite_amd import batch
    N = 1000
    kernel = terms.RealTerm(log_a=0.1, log_c=0.5) + terms.SHOTerm(log_S0=0.1, log_Q=1.0, log_omega0=0.5)
    assert kernel.coefficients[0].size == 1 and kernel.coefficients[2].size == 1   # width 3: Q >= 1/2 -> one complex pair
    rng = np.random.RandomState(1000)
    t = np.sort(rng.uniform(0, 50, N))
    yerr = rng.uniform(0.1, 0.3, N)
    y = np.sin(t) + yerr * rng.randn(N)
    gp = GP(kernel)
    gp.compute(t, yerr)
    ll = gp.log_likelihood(y)
    r = ref.RefSolver()
    r.compute(kernel.jitter, *kernel.coefficients, *NO_GENERAL, t, yerr ** 2)
    ld0, q0 = r.log_determinant(), r.dot_solve(y)
    ll0 = -0.5 * (q0 + ld0 + N * np.log(2.0 * np.pi))
    assert rel(gp.solver.log_determinant(), ld0) <= REL
    assert rel(gp.solver.dot_solve(y), q0) <= REL
    assert rel(ll, ll0) <= REL
    # a second draw of the hyper-parameters through the same GP (recompute-if-dirty, celerite.py:160-178)
    p = gp.get_parameter_vector()
    gp.set_parameter_vector(p + 0.05)
    r.compute(kernel.jitter, *kernel.coefficients, *NO_GENERAL, t, yerr ** 2)
    ll1 = -0.5 * (r.dot_solve(y) + r.log_determinant() + N * np.log(2.0 * np.pi))
    assert rel(gp.log_likelihood(y), ll1) <= REL
    # the batched entry over a table of draws: every draw against the oracle
    draws = p[None, :] + 0.05 * rng.randn(8, len(p))
    tab = batch.kernel_coefficient_table(kernel, draws)
    llb, ldb, qb, st = batch.batch_log_likelihood(*tab[:6], t, yerr ** 2, y, jitter=tab[6])
    assert (st == 0).all()
    for i in range(len(draws)):
        gp.set_parameter_vector(draws[i])
        r.compute(kernel.jitter, *kernel.coefficients, *NO_GENERAL, t, yerr ** 2)
        assert rel(ldb[i], r.log_determinant()) <= REL and rel(qb[i], r.dot_solve(y)) <= REL


# ---- golden value and 1e-10 parity -----------------------------------------------
def test_first_tutorial_known_answer():  # docs/tutorials/first.rst:24-31,74-101
    t, yerr, y = first_tutorial_case()
    Q = 1.0 / np.sqrt(2.0)
    w0 = 3.0
    S0 = np.var(y) / (w0 * Q)
    bounds = dict(log_S0=(-15, 15), log_Q=(-15, 15), log_omega0=(-15, 15))
    kernel = terms.SHOTerm(log_S0=np.log(S0), log_Q=np.log(Q), log_omega0=np.log(w0), bounds=bounds)
    kernel.freeze_parameter("log_Q")
    Q = 1.0
    S0 = np.var(y) / (w0 * Q)
    kernel += terms.SHOTerm(log_S0=np.log(S0), log_Q=np.log(Q), log_omega0=np.log(w0), bounds=bounds)
    gp = celerite_amd.GP(kernel, mean=np.mean(y))
    gp.compute(t, yerr)
    assert abs(gp.log_likelihood(y) - FIRST_TUTORIAL_LOGLIKE) <= 1e-10 * abs(FIRST_TUTORIAL_LOGLIKE)
    assert gp.get_parameter_names() == ("kernel:terms[0]:log_S0", "kernel:terms[0]:log_omega0",
                                        "kernel:terms[1]:log_S0", "kernel:terms[1]:log_Q",
                                        "kernel:terms[1]:log_omega0")


def test_cc_suite_1e10():
    """cpp/src/test_solvers.cc:79-97: real / complex / mixed / general kernels, N = 1024,
    jitter 0.01; there the bar is 1e-10 ABSOLUTE against a dense LDL^T."""
    rng = np.random.RandomState(42)
    N = 1024
    x = np.sort(rng.uniform(-1, 1, N))
    yerr2 = 0.3 + 0.1 * rng.uniform(-1, 1, N)
    y = np.sin(x)
    U = np.array([x ** j for j in range(3)])
    V = np.array([x ** j / (1.0 + j) for j in range(3)])
    A = 1e-6 + np.sum(U * V, axis=0)
    e = np.empty(0)
    for coeffs, gen in [(COEFFS_CC_REAL + (e, e, e, e), NO_GENERAL), ((e, e) + COEFFS_CC_COMP, NO_GENERAL),
                        (COEFFS_CC_REAL + COEFFS_CC_COMP, NO_GENERAL),
                        (COEFFS_CC_REAL + COEFFS_CC_COMP, (A, U, V))]:
        s = celerite_amd.CholeskySolver()
        s.compute(0.01, *coeffs, *gen, x, yerr2)
        K = dense.dense_matrix(0.01, *coeffs, *gen, x, yerr2)
        assert abs(s.log_determinant() - dense.dense_logdet(K)) < 1e-10
        assert abs(s.dot_solve(y) - y @ np.linalg.solve(K, y)) < 1e-10


@pytest.mark.parametrize("N", [1, 2, 3, 127, 128, 129, 1000, 20000])
@pytest.mark.parametrize("shape", [(1, 0), (1, 1), (2, 3), (8, 0), (0, 4)])
def test_solver_vs_oracle_small_widths(N, shape):
    """Widths <= 8 take the chunked-scan path inside compute(); N straddles the
    single-chunk / multi-chunk switch."""
    case = synthetic(1, N, shape[0], shape[1], "accuracy" if N % 2 else "bench", seed=N)
    co = coeffs_of(case, 0)
    t, diag, y = case["t"][0], case["diag"][0], case["y"][0]
    s, r = celerite_amd.CholeskySolver(), ref.RefSolver()
    s.compute(0.0, *co, *NO_GENERAL, t, diag)
    r.compute(0.0, *co, *NO_GENERAL, t, diag)
    assert rel(s.log_determinant(), r.log_determinant()) <= REL
    assert rel(s.dot_solve(y), r.dot_solve(y)) <= REL
    xs, xr = s.solve(y), r.solve(y)
    assert np.abs(xs - xr).max() <= 1e-9 * np.abs(xr).max()


def test_wide_and_jitter_only_kernels_vs_oracle():
    golden = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden",
                                         "terms_golden.json")))
    co = [np.array(b) for b in golden["test_log_likelihood full kernel (width 26)"]["coefficients"]]
    np.random.seed(1)
    t = np.sort(np.random.rand(300))
    diag = np.random.uniform(0.1, 0.5, 300)
    y = np.random.randn(300)
    gen = general_terms(t, np.random.rand)
    for g in (NO_GENERAL, gen):  # widths 26 and 30
        s, r = celerite_amd.CholeskySolver(), ref.RefSolver()
        s.compute(0.01, *co, *g, t, diag)
        r.compute(0.01, *co, *g, t, diag)
        assert rel(s.log_determinant(), r.log_determinant()) <= REL
        assert rel(s.dot_solve(y), r.dot_solve(y)) <= REL
    e = np.empty(0)
    s, r = celerite_amd.CholeskySolver(), ref.RefSolver()  # J == 0: cholesky.h:90-95
    s.compute(0.3, e, e, e, e, e, e, *NO_GENERAL, t, diag)
    r.compute(0.3, e, e, e, e, e, e, *NO_GENERAL, t, diag)
    assert rel(s.log_determinant(), r.log_determinant()) <= REL
    assert rel(s.dot_solve(y), r.dot_solve(y)) <= REL
    assert np.allclose(s.solve(y)[:, 0], y / (diag + 0.3))
    assert np.allclose(s.dot(0.3, e, e, e, e, e, e, *NO_GENERAL, t, y)[:, 0], 0.3 * y)


def test_failed_compute_leaves_solver_not_computed():  # cholesky.h:57
    e = np.empty(0)
    s = celerite_amd.CholeskySolver()
    t = np.linspace(0, 1, 50)
    s.compute(0.0, np.ones(1), np.ones(1), e, e, e, e, *NO_GENERAL, t, np.ones(50))
    assert s.computed()
    with pytest.raises(LinAlgError):
        s.compute(0.0, -np.ones(1), 0.1 * np.ones(1), e, e, e, e, *NO_GENERAL, t, np.zeros(50))
    assert not s.computed()
    with pytest.raises(RuntimeError):
        s.log_determinant()
    with pytest.raises(RuntimeError, match="dimension mismatch"):
        s.compute(0.0, np.ones(1), np.ones(1), e, e, e, e, *NO_GENERAL, t, np.ones(49))


@pytest.mark.parametrize("JR,JC,N", [(1, 4, 300), (2, 7, 3000), (4, 11, 5000), (0, 16, 20000), (6, 13, 2500), (10, 15, 700),
                                     (8, 16, 30000), (4, 22, 12000), (0, 32, 20000), (64, 0, 9000)])
def test_object_api_wide_widths_through_the_wide_scan(JR, JC, N):
    """CholeskySolver.compute at widths 9..64 without general terms runs the batched wide kernels on
    one problem (chunked for N >= 2048; round 5: widths 33..64 too -- sqrt(0.011 N) chunks chained by one walk,
    csrc/wide64_kernels.hip) and writes the factor in the reference's
    storage: log_determinant, dot_solve, solve and the pickled state against the oracle."""
    rng = np.random.RandomState(JR * 100 + JC)
    t = np.sort(rng.uniform(0, 0.05 * N, N))
    yerr = rng.uniform(0.3, 0.5, N)
    y = rng.randn(N)
    args = (0.0, np.exp(rng.uniform(-1, 0.5, JR)), np.exp(rng.uniform(-2, 0, JR)), np.exp(rng.uniform(-1, 0.5, JC)),
            0.1 * rng.rand(JC), np.exp(rng.uniform(-2, 0, JC)), np.exp(rng.uniform(0, 3, JC)),
            np.empty(0), np.empty((0, 0)), np.empty((0, 0)), t, yerr ** 2)
    s = celerite_amd.CholeskySolver()
    r = ref.RefSolver()
    s.compute(*args)
    r.compute(*args)
    assert abs(s.log_determinant() - r.log_determinant()) <= 1e-10 * abs(r.log_determinant())
    assert abs(s.dot_solve(y) - r.dot_solve(y)) <= 1e-10 * abs(r.dot_solve(y))
    assert np.allclose(s.solve(y), r.solve(y), rtol=1e-9, atol=1e-12)
    st, st0 = s.__getstate__(), r.state()
    assert st[:3] == (True, N, JR + 2 * JC)
    for a, b in zip(st[4:], st0[4:]):
        assert np.allclose(a, b, rtol=1e-9, atol=1e-13)


@pytest.mark.parametrize("JR,JC", [(2, 5), (0, 8), (4, 11), (0, 16)])
def test_object_api_wide_widths_around_the_chunking_thresholds(JR, JC):
    """Widths 9..32 through `CholeskySolver`: from N = 256 (width <= 16) / 512 on the series is cut into >= 8 chunks of
    >= 48 / 96 samples and the chunks' prefix is a parallel scan (csrc/wide_prefix_scan.hip); below, one sequential sweep.
    Both sides of each threshold, ragged last chunks, and the stored factor must equal the oracle's."""
    for N in (255, 256, 257, 383, 385, 511, 512, 513, 767, 1100, 3100):
        rng = np.random.RandomState(N + JR)
        t = np.sort(rng.uniform(0, 0.05 * N, N))
        yerr = rng.uniform(0.3, 0.5, N)
        y = rng.randn(N)
        args = (0.0, np.exp(rng.uniform(-1, 0.5, JR)), np.exp(rng.uniform(-2, 0, JR)), np.exp(rng.uniform(-1, 0.5, JC)),
                0.1 * rng.rand(JC), np.exp(rng.uniform(-2, 0, JC)), np.exp(rng.uniform(0, 3, JC)),
                np.empty(0), np.empty((0, 0)), np.empty((0, 0)), t, yerr ** 2)
        s, r = celerite_amd.CholeskySolver(), ref.RefSolver()
        try:
            r.compute(*args)
        except Exception:      # (a random complex term with b != 0 need not be positive definite: same verdict wanted)
            with pytest.raises(celerite_amd.solver.LinAlgError):
                s.compute(*args)
            continue
        s.compute(*args)
        assert abs(s.log_determinant() - r.log_determinant()) <= 1e-10 * abs(r.log_determinant()), N
        assert abs(s.dot_solve(y) - r.dot_solve(y)) <= 1e-10 * abs(r.dot_solve(y)), N
        st, st0 = s.__getstate__(), r.state()
        assert st[:3] == (True, N, JR + 2 * JC)
        for a, b in zip(st[4:], st0[4:]):
            assert np.allclose(a, b, rtol=1e-9, atol=1e-13), N


@pytest.mark.parametrize("JR,JC,N", [(1, 1, 900), (2, 3, 6000), (2, 7, 5000), (0, 16, 4000)])
def test_hinted_right_hand_side_is_the_same_quadratic_form(JR, JC, N):
    """GP.log_likelihood announces its residual before the factorisation (solver._hint_rhs): compute then
    returns resid^T K^-1 resid from its own pass.  Same value as the ordinary dot_solve sweep, only for
    that very vector, and only once."""
    rng = np.random.RandomState(JR * 10 + JC)
    t = np.sort(rng.uniform(0, 0.05 * N, N))
    yerr = rng.uniform(0.3, 0.5, N)
    y, y2 = rng.randn(N), rng.randn(N)
    args = (0.0, np.exp(rng.uniform(-1, 0.5, JR)), np.exp(rng.uniform(-2, 0, JR)), np.exp(rng.uniform(-1, 0.5, JC)),
            np.zeros(JC), np.exp(rng.uniform(-2, 0, JC)), np.exp(rng.uniform(0, 3, JC)),
            np.empty(0), np.empty((0, 0)), np.empty((0, 0)), t, yerr ** 2)
    r = ref.RefSolver()
    r.compute(*args)
    s = celerite_amd.CholeskySolver()
    s._hint_rhs(y)
    s.compute(*args)
    assert abs(s.dot_solve(y) - r.dot_solve(y)) <= 1e-10 * abs(r.dot_solve(y))        # from compute's pass
    assert abs(s.dot_solve(y2) - r.dot_solve(y2)) <= 1e-10 * abs(r.dot_solve(y2))     # ordinary sweep
    assert abs(s.log_determinant() - r.log_determinant()) <= 1e-10 * abs(r.log_determinant())
    s.compute(*args)                                                                  # the hint is one-shot
    assert abs(s.dot_solve(y) - r.dot_solve(y)) <= 1e-10 * abs(r.dot_solve(y))
    # through GP: log_likelihood hints by itself
    k = terms.RealTerm(log_a=0.1, log_c=-1.0)
    for j in range(JC):
        k += terms.ComplexTerm(log_a=-0.5, log_c=-1.0 - 0.1 * j, log_d=0.3 * j)
    gp = GP(k)
    gp.compute(t, yerr)
    ll1 = gp.log_likelihood(y)
    gp.set_parameter_vector(gp.get_parameter_vector())   # dirty -> recompute with the hint
    ll2 = gp.log_likelihood(y)
    assert abs(ll1 - ll2) <= 1e-10 * abs(ll1)


@pytest.mark.parametrize("JR,JC,N,general", [(1, 4, 2048, False), (3, 5, 4097, True), (0, 8, 30000, False),
                                              (2, 15, 12345, False), (0, 16, 100000, False), (9, 0, 5000, False),
                                              (2, 20, 6000, False),
                                              # round 3: widths 33..64 (incl. general terms: 37 + 3 = 40, 61 + 3 = 64),
                                              # the 65-row chunk map of width 64, short wide series (N >= 512)
                                              (0, 20, 6000, False), (3, 17, 3000, True), (4, 22, 9000, False),
                                              (0, 28, 2500, False), (1, 30, 4000, True), (0, 32, 5000, False),
                                              (63, 0, 2100, False), (3, 13, 600, False), (2, 30, 777, False),
                                              (0, 5, 512, False)])
def test_wide_sweeps_are_chunked_scans(JR, JC, N, general):
    """dot_solve / solve / dot_L (widths 9..64) on a stored factor run as chunked scans (csrc/wsweep_kernels.hip:
    lane = column of the chunk's affine map / lane = row) from N = 2048 (N = 512 above width 8) on:
    same numbers as the oracle's sequential sweeps (cholesky.h:218-431), several right-hand sides."""
    rng = np.random.RandomState(JR * 100 + JC + N % 7)
    t = np.sort(rng.uniform(0, 0.05 * N, N))
    diag = rng.uniform(0.1, 0.3, N)
    gen = NO_GENERAL
    if general:
        U = np.vander((t - t.mean()) / (t.max() - t.min()), 3).T
        V = U * rng.rand(3)[:, None]
        gen = (np.sum(U * V, axis=0) + 1e-8, U, V)
    co = (np.exp(rng.uniform(-1, 0.5, JR)), np.exp(rng.uniform(-2, 0, JR)), np.exp(rng.uniform(-1, 0.5, JC)),
          0.1 * rng.rand(JC), np.exp(rng.uniform(-2, 0, JC)), np.exp(rng.uniform(0, 3, JC)))
    s = celerite_amd.CholeskySolver()
    r = ref.RefSolver()
    s.compute(0.0, *co, *gen, t, diag)
    r.compute(0.0, *co, *gen, t, diag)
    b = rng.randn(N, 3)
    for k in range(2):
        q, q0 = s.dot_solve(b[:, k]), r.dot_solve(b[:, k])
        assert abs(q - q0) <= 1e-10 * abs(q0)
    x, x0 = s.solve(b), r.solve(b)
    assert x.shape == (N, 3)
    assert np.max(np.abs(x - x0)) <= 1e-10 * np.max(np.abs(x0))
    yl, yl0 = s.dot_L(b), r.dot_L(b)
    assert np.max(np.abs(yl - yl0)) <= 1e-11 * np.max(np.abs(yl0))
    if not general:
        y = np.sin(t) + 0.1 * b[:, 0]
        xs = np.linspace(t[0] - 1.0, t[-1] + 1.0, 333)
        p, p0 = s.predict(y, xs), r.predict(y, xs)
        assert np.max(np.abs(p - p0)) <= 1e-9 * max(1.0, np.max(np.abs(p0)))


def test_carma():  # tests/test_celerite.py:22-42
    """The reference's own CARMA test: the Kalman-filter log-likelihood (csrc/carma.hip) equals the celerite
    log-likelihood of get_celerite_coeffs() through CholeskySolver."""
    solver = celerite_amd.CholeskySolver()
    np.random.seed(42)
    t = np.sort(np.random.uniform(0, 5, 100))
    yerr = 0.1 + np.zeros_like(t)
    y = np.sin(t) + yerr * np.random.randn(len(t))
    carma_solver = celerite_amd.solver.CARMASolver(-0.5, np.array([0.1, 0.05, 0.01]), np.array([0.2, 0.1]))
    carma_ll = carma_solver.log_likelihood(t, y, yerr)
    params = carma_solver.get_celerite_coeffs()
    solver.compute(0.0, params[0], params[1], params[2], params[3], params[4], params[5],
                   np.empty(0), np.empty((0, 0)), np.empty((0, 0)), t, yerr ** 2)
    celerite_ll = -0.5 * (solver.dot_solve(y) + solver.log_determinant() + len(t) * np.log(2 * np.pi))
    assert np.allclose(carma_ll, celerite_ll)
    assert abs(carma_ll - celerite_ll) <= 1e-10 * abs(celerite_ll)


@pytest.mark.parametrize("log_sigma,ar,ma,N", [(-0.5, [0.1, 0.05, 0.01], [0.2, 0.1], 100), (0.3, [0.5, -0.2], [0.1], 777),
                                               (0.0, [1.0, 0.3, -0.4, 0.2, 0.05], [0.3, -0.1, 0.2], 2000),
                                               (-1.0, [0.4], [], 1), (0.2, [0.3, 0.1, -0.2, 0.4, 0.0, 0.2, -0.1, 0.3, 0.1], [0.1, 0.2], 300)])
def test_carma_filter_vs_oracle(log_sigma, ar, ma, N):
    """CARMASolver.log_likelihood on the device against oracle/carma.py (carma.h:221-239 restated) at 1e-10,
    orders 1..9, irregular sampling, heteroscedastic errors; repeated calls on one object."""
    from oracle import carma
    rng = np.random.RandomState(N)
    t = np.sort(rng.uniform(0, 0.05 * N + 1, N))
    yerr = rng.uniform(0.05, 0.3, N)
    y = np.sin(t) + yerr * rng.randn(N)
    s = celerite_amd.solver.CARMASolver(log_sigma, np.array(ar), np.array(ma, dtype=float))
    o = carma.CARMASolver(log_sigma, ar, ma)
    want = o.log_likelihood(t, y, yerr)
    for _ in range(2):
        got = s.log_likelihood(t, y, yerr)
        assert abs(got - want) <= 1e-10 * abs(want)
    got2 = s.log_likelihood(t[: N // 2 + 1], y[: N // 2 + 1], yerr[: N // 2 + 1])
    want2 = o.log_likelihood(t[: N // 2 + 1], y[: N // 2 + 1], yerr[: N // 2 + 1])
    assert abs(got2 - want2) <= 1e-10 * abs(want2)
    with pytest.raises(RuntimeError, match="dimension mismatch"):      # carma.h:223
        s.log_likelihood(t, y[:-1], yerr)


def test_carma_instability_raises():  # carma.h:185-186, exceptions.h:8-12
    """A predicted variance below zero: the reference throws carma_exception (-> RuntimeError).  Unsorted times
    (negative dt) make the propagators exceed one and drive P indefinite; the oracle raises at the same input."""
    from oracle import carma
    ar, ma = [0.1, 0.05, 0.01], [0.2, 0.1]
    s = celerite_amd.solver.CARMASolver(-0.5, np.array(ar), np.array(ma))
    o = carma.CARMASolver(-0.5, ar, ma)
    t_bad = np.array([0.0, 1.0, 0.5, 3.0, 0.0])
    with pytest.raises(carma.CarmaInstability):
        o.log_likelihood(t_bad, np.zeros(5), np.zeros(5))
    with pytest.raises(RuntimeError, match="CARMA model encountered an instability"):
        s.log_likelihood(t_bad, np.zeros(5), np.zeros(5))
    t = np.linspace(0, 1, 5)                                            # the object stays usable
    got, want = s.log_likelihood(t, np.ones(5), 0.1 + np.zeros(5)), o.log_likelihood(t, np.ones(5), 0.1 + np.zeros(5))
    assert abs(got - want) <= 1e-10 * abs(want)


@pytest.mark.parametrize("JR,JC,nrhs,N", [(0, 16, 40, 3000), (2, 3, 70, 3000), (1, 7, 64, 3000),
                                          (2, 3, 21, 17000), (0, 16, 9, 20001), (1, 12, 3, 16384)])
def test_wide_sweeps_many_right_hand_sides(JR, JC, nrhs, N):
    """solve with more right-hand sides than one wave of columns holds (the chunk map's J + 1 columns plus nrhs
    affine columns spill into further column blocks of wsweep_summarize_kernel; the prefix and replay run one
    workgroup / wave per right-hand side).  From N = 16384 on the prefix runs in two levels: the run maps are composed
    in blocks of 8 columns of [P | a_1 .. a_nrhs] (wsweep_compose_kernel) -- several blocks here."""
    rng = np.random.RandomState(nrhs)
    t = np.sort(rng.uniform(0, 0.05 * N, N))
    diag = rng.uniform(0.1, 0.3, N)
    co = (np.exp(rng.uniform(-1, 0.5, JR)), np.exp(rng.uniform(-2, 0, JR)), np.exp(rng.uniform(-1, 0.5, JC)),
          0.1 * rng.rand(JC), np.exp(rng.uniform(-2, 0, JC)), np.exp(rng.uniform(0, 3, JC)))
    s, r = celerite_amd.CholeskySolver(), ref.RefSolver()
    s.compute(0.0, *co, *NO_GENERAL, t, diag)
    r.compute(0.0, *co, *NO_GENERAL, t, diag)
    b = rng.randn(N, nrhs)
    x, x0 = s.solve(b), r.solve(b)
    assert x.shape == (N, nrhs)
    assert np.max(np.abs(x - x0)) <= 1e-10 * np.max(np.abs(x0))
    yl, yl0 = s.dot_L(b), r.dot_L(b)
    assert np.max(np.abs(yl - yl0)) <= 1e-11 * np.max(np.abs(yl0))


@pytest.mark.parametrize("JR,JC,N,general", [(1, 1, 2500, False), (2, 3, 6000, True), (0, 8, 4000, False), (3, 14, 3000, True)])
def test_dot_long_series_is_a_chunked_scan(JR, JC, N, general):
    """dot (cholesky.h:444-590) at N >= 2048: both triangles as chunked diagonal scans (wdot_kernel in
    csrc/wsweep_kernels.hip): against a dense K z and against the oracle's sequential sweeps."""
    rng = np.random.RandomState(JR * 10 + JC)
    t = np.sort(rng.uniform(0, 0.05 * N, N))
    co = (np.exp(rng.uniform(-1, 0.5, JR)), np.exp(rng.uniform(-2, 0, JR)), np.exp(rng.uniform(-1, 0.5, JC)),
          0.1 * rng.rand(JC), np.exp(rng.uniform(-2, 0, JC)), np.exp(rng.uniform(0, 3, JC)))
    gen = NO_GENERAL
    if general:
        U = np.vander((t - t.mean()) / (t.max() - t.min()), 3).T
        V = U * rng.rand(3)[:, None]
        gen = (np.sum(U * V, axis=0) + 1e-8, U, V)
    z = rng.randn(N, 3)
    s, r = celerite_amd.CholeskySolver(), ref.RefSolver()
    y = s.dot(0.3, *co, *gen, t, z)
    y0 = r.dot(0.3, *co, *gen, t, z)
    assert y.shape == (N, 3)
    assert np.max(np.abs(y - y0)) <= 1e-11 * np.max(np.abs(y0))
    K = get_kernel_value(*co, t[:, None] - t[None, :])
    K[np.diag_indices_from(K)] += 0.3
    if general:
        K[np.diag_indices_from(K)] += gen[0]
        K += np.tril(np.dot(gen[1].T, gen[2]), -1) + np.triu(np.dot(gen[2].T, gen[1]), 1)
    assert np.allclose(np.dot(K, z), y, rtol=1e-9, atol=1e-9)


@pytest.mark.parametrize("JR,JC,spread", [(2, 3, False), (0, 8, True), (0, 16, True)])
def test_long_series_factor_and_solve_are_as_accurate_as_the_sequential_recurrence(JR, JC, spread):
    """Round 6: one series of 1e5 samples through ``CholeskySolver`` (the reference's own API, cholesky.h:41-318).  The
    factor is written chunk by chunk from SCANNED start states whose rounding showed in W and D at the chunk heads -- with
    ~50-sample chunks that was most of the factor: W 1.1e-10, ``solve`` 1.8e-10 of the oracle at width 8
    (profiles/r06n_solver_factor_error.txt).  The first call that reads the factor now replays the chunks once more from
    the previous chunk's replayed end state: the state (W, D), ``solve``, ``dot_solve`` of a fresh vector, ``dot_L`` and
    ``predict`` within 2e-11 of the oracle, in whatever order they are first called; ``log_determinant`` and the hinted
    ``dot_solve`` (GP.log_likelihood) are what they were; a pickled solver carries the refined factor."""
    from bench import make_inputs
    N = 100000
    coeffs, t, diag, y = make_inputs(1, N, JR, JC, 42, d_spread=spread)
    cs = [c[0] for c in coeffs]
    e_, e2_ = np.empty(0), np.empty((0, 0))
    r = ref.RefSolver()
    r.compute(0.0, *cs, e_, e2_, e2_, t[0], diag[0])
    _, _, J, logdet, rphi, ru, rW, rD = r.state()
    rng = np.random.RandomState(5)
    z = rng.randn(N)
    xs = np.sort(rng.uniform(t[0].min(), t[0].max(), 500))
    want = {"solve": r.solve(y[0])[:, 0], "dot_solve": r.dot_solve(z), "dot_L": r.dot_L(z)[:, 0], "predict": r.predict(y[0], xs)}
    calls = {"solve": lambda s: s.solve(y[0])[:, 0], "dot_solve": lambda s: s.dot_solve(z), "dot_L": lambda s: s.dot_L(z)[:, 0],
             "predict": lambda s: s.predict(y[0], xs)}
    for first in ("solve", "dot_L", "predict", "dot_solve", "state"):
        s = celerite_amd.CholeskySolver()
        s._hint_rhs(y[0])
        s.compute(0.0, *cs, e_, e2_, e2_, t[0], diag[0])
        assert abs(s.log_determinant() - logdet) <= 1e-12 * abs(logdet)
        assert abs(s.dot_solve(y[0]) - r.dot_solve(y[0])) <= 1e-11 * abs(r.dot_solve(y[0]))    # (the hinted vector: no factor read)
        order = [first] + [k for k in calls if k != first] if first != "state" else list(calls)
        if first == "state":
            s = pickle.loads(pickle.dumps(s, -1))
        for k in order:
            got = np.atleast_1d(calls[k](s)) if (first != "state" or k != "predict") else None
            if got is None:
                continue                                   # (a restored solver cannot predict: solver.cpp:36-42)
            w = np.atleast_1d(want[k])
            within("one long series (N = 1e5) through CholeskySolver, %s vs oracle (of the largest)" % k,
                   np.max(np.abs(got - w)) / np.max(np.abs(w)), 2e-11, (JR, JC, first))
        st = s.__getstate__()
        W, D = np.asarray(st[6]).reshape(rW.shape), np.asarray(st[7])
        within("one long series (N = 1e5) through CholeskySolver, state: W vs oracle (of the largest entry)", np.max(np.abs(W - rW)) / np.max(np.abs(rW)), 1e-11, (JR, JC, first))
        within("one long series (N = 1e5) through CholeskySolver, state: D vs oracle (relative)", np.max(np.abs(D - rD) / np.abs(rD)), 1e-11, (JR, JC, first))


@pytest.mark.parametrize("JR,JC,N", [(1, 64, 600), (0, 100, 400), (3, 254, 300), (130, 0, 500), (2, 70, 200)])
def test_any_width_above_128_through_the_object_api(JR, JC, N):
    """Round 6 (VERDICT r5 missing #5, second half): the reference's dynamic-width arm takes ANY J
    (``FIXED_SIZE_HACKZ(Eigen::Dynamic)``, cholesky.h:203; its benchmark goes to width 512, examples/benchmark/run.py:39).
    ``CholeskySolver.compute / log_determinant / dot_solve / solve`` now do too, up to ``CLR_MAX_WIDTH_ANY`` = 1024
    (csrc/huge_kernels.hip: S in HBM / L2, one workgroup walks the series): against the oracle at widths 129, 200, 511
    and 130 real terms, the hinted and the plain ``dot_solve``, several right-hand sides, the pickled state, a problem
    that is not positive definite; (later in round 6) ``dot_L``, ``dot`` and ``predict`` above 128 too."""
    J = JR + 2 * JC
    assert J > 128
    case = synthetic(1, N, JR, JC, "accuracy", seed=J)
    cs = list(coeffs_of(case, 0))
    t, diag, y = case["t"][0], case["diag"][0] + 0.05, case["y"][0]
    r = ref.RefSolver()
    r.compute(0.1, *cs, *NO_GENERAL, t, diag)
    rng = np.random.RandomState(J)
    b = rng.randn(N, 3)
    s = celerite_amd.CholeskySolver()
    s._hint_rhs(y)
    s.compute(0.1, *cs, *NO_GENERAL, t, diag)
    assert s.computed()
    within("widths above 128 (object API): log det vs oracle", abs(s.log_determinant() - r.log_determinant()) / abs(r.log_determinant()), 1e-10, J)
    within("widths above 128 (object API): hinted dot_solve vs oracle", abs(s.dot_solve(y) - r.dot_solve(y)) / abs(r.dot_solve(y)), 1e-10, J)
    within("widths above 128 (object API): dot_solve vs oracle", abs(s.dot_solve(b[:, 1]) - r.dot_solve(b[:, 1])) / abs(r.dot_solve(b[:, 1])), 1e-10, J)
    want = r.solve(b)
    got = s.solve(b)
    assert got.shape == (N, 3)
    within("widths above 128 (object API): solve vs oracle (of the largest entry)", np.max(np.abs(got - want)) / np.max(np.abs(want)), 1e-10, J)
    s2 = pickle.loads(pickle.dumps(s, -1))
    assert np.array_equal(s2.solve(b), got) and s2.log_determinant() == s.log_determinant()
    # round 6, later: dot_L (the diagonal scans with one thread per row, 2 .. 16 waves), dot and predict (sorted points:
    # the scans; unsorted: the sequential walk of short series with up to 16 rows per lane) above width 128 as well
    within("widths above 128 (object API): dot_L vs oracle (of the largest entry)", np.max(np.abs(s.dot_L(b) - r.dot_L(b))) / np.max(np.abs(r.dot_L(b))), 1e-10, J)
    want_dot = r.dot(0.1, *cs, *NO_GENERAL, t, b)
    got_dot = celerite_amd.CholeskySolver().dot(0.1, *cs, *NO_GENERAL, t, b)
    within("widths above 128 (object API): dot vs oracle (of the largest entry)", np.max(np.abs(got_dot - want_dot)) / np.max(np.abs(want_dot)), 1e-10, J)
    rng2 = np.random.RandomState(J + 1)
    xs = rng2.uniform(t.min() - 0.5, t.max() + 0.5, 40)
    pts = np.sort(xs)                  # (sorted, as the reference's walk assumes: N >= 256 the scans, below the sequential walk)
    want_p = r.predict(y, pts)
    within("widths above 128 (object API): predict vs oracle (of the largest entry)", np.max(np.abs(s.predict(y, pts) - want_p)) / np.max(np.abs(want_p)), 1e-10, J)
    bad = list(cs)
    if JR:
        bad[0] = -20.0 * np.abs(bad[0])
    else:
        bad[2] = -20.0 * np.abs(bad[2])
    s3 = celerite_amd.CholeskySolver()
    with pytest.raises(LinAlgError):
        s3.compute(0.0, *bad, *NO_GENERAL, t, np.zeros(N))
    assert not s3.computed()
    with pytest.raises(RuntimeError):        # ... and above CLR_MAX_WIDTH_ANY: refused
        celerite_amd.CholeskySolver().compute(0.0, np.ones(1100), np.ones(1100), E0, E0, E0, E0, *NO_GENERAL, t, diag)


def _reference_benchmark_kernel(width):
    """The kernels of the reference's own benchmark (examples/benchmark/run.py:80-84): real terms (1.0, 0.1) and IDENTICAL
    complex terms (0.1, 2.0, 1.6) up to the width asked for."""
    j = width // 2
    kernel = terms.RealTerm(1.0, 0.1)
    for _ in range((2 * j - 1) % 2):
        kernel += terms.RealTerm(1.0, 0.1)
    for _ in range((2 * j - 1) // 2):
        kernel += terms.ComplexTerm(0.1, 2.0, 1.6)
    return kernel


@pytest.mark.parametrize("width,N", [(16, 8192), (16, 65536), (32, 8192), (32, 65536)])
def test_reference_benchmark_kernels_are_settled_by_the_output_check(width, N):
    """Round 6: identical terms leave directions of the state no sample ever probes; the rounding of the scan and of the
    recurrence collects there, the chunked replay's END STATES miss the scanned start states by 1.2e-11 .. 2.8e-11 (bound
    1e-11) and ``CholeskySolver.compute`` used to hand the reference's own benchmark problems to the sequential recurrence
    (width 16, N = 65536: 29.6 ms against the CPU's 11.2; profiles/r06q_family_factor.txt).  Now the chunks are replayed
    again from the previous replay's end states until what two consecutive replays WROTE agrees to 2e-11
    (BatchParams::head_check): level 1, the factor within the sequential recurrence's own distance of the oracle -- and
    with the check switched off the same problem still takes the sequential route and gives the same answers."""
    rng = np.random.RandomState(42)
    t = np.sort(rng.rand(2 ** 19))[:N]
    yerr = rng.uniform(0.1, 0.2, 2 ** 19)[:N]
    y = np.sin(t)
    cs = [np.asarray(c, dtype=float) for c in _reference_benchmark_kernel(width).coefficients]
    e_, e2_ = np.empty(0), np.empty((0, 0))
    r = ref.RefSolver()
    r.compute(0.0, *cs, e_, e2_, e2_, t, yerr ** 2)
    _, _, J, logdet, rphi, ru, rW, rD = r.state()
    assert J == width
    want_solve, want_quad = r.solve(y)[:, 0], r.dot_solve(y)
    got = {}
    for check in (True, False):
        batch.set_option("CLR_OUTPUT_CHECK_CAP", None if check else "0")
        try:
            s = celerite_amd.CholeskySolver()
            s._hint_rhs(y)
            s.compute(0.0, *cs, e_, e2_, e2_, t, yerr ** 2)
            level, nchunk, residual = s._route()
        finally:
            batch.set_option("CLR_OUTPUT_CHECK_CAP", None)
        assert nchunk > 1
        if check:
            assert level == 1 and residual <= 2e-11, (level, residual)    # (settled by consecutive replays that agree)
        else:
            assert level == 2 and residual > 1e-11, (level, residual)     # (the end-state test alone: sequential)
        tag = (width, N, "output check" if check else "end-state test only")
        within("reference benchmark kernels through CholeskySolver: log det vs oracle", abs(s.log_determinant() - logdet) / abs(logdet), 1e-12, tag)
        within("reference benchmark kernels through CholeskySolver: hinted dot_solve vs oracle", abs(s.dot_solve(y) - want_quad) / abs(want_quad), 1e-11, tag)
        sv = s.solve(y)[:, 0]
        within("reference benchmark kernels through CholeskySolver: solve vs oracle (of the largest)", np.max(np.abs(sv - want_solve)) / np.max(np.abs(want_solve)), 2e-11, tag)
        st = s.__getstate__()
        W, D = np.asarray(st[6]).reshape(rW.shape), np.asarray(st[7])
        within("reference benchmark kernels through CholeskySolver: W vs oracle (of the largest entry)", np.max(np.abs(W - rW)) / np.max(np.abs(rW)), 2e-11, tag)
        within("reference benchmark kernels through CholeskySolver: D vs oracle (relative)", np.max(np.abs(D - rD) / np.abs(rD)), 3e-11, tag)
        got[check] = (s.log_determinant(), sv)
    assert abs(got[True][0] - got[False][0]) <= 1e-12 * abs(logdet)


def test_output_check_that_never_agrees_ends_in_the_sequential_recurrence():
    """The other exit of the output check: with an unreachable tolerance every attempt fails and the problem is settled by
    the sequential recurrence -- same answers, level 2."""
    N, width = 8192, 16
    rng = np.random.RandomState(42)
    t = np.sort(rng.rand(2 ** 19))[:N]
    yerr = rng.uniform(0.1, 0.2, 2 ** 19)[:N]
    cs = [np.asarray(c, dtype=float) for c in _reference_benchmark_kernel(width).coefficients]
    e_, e2_ = np.empty(0), np.empty((0, 0))
    r = ref.RefSolver()
    r.compute(0.0, *cs, e_, e2_, e2_, t, yerr ** 2)
    batch.set_option("CLR_OUTPUT_CHECK_TOL", "1e-300")
    try:
        s = celerite_amd.CholeskySolver()
        s.compute(0.0, *cs, e_, e2_, e2_, t, yerr ** 2)
        level, nchunk, residual = s._route()
    finally:
        batch.set_option("CLR_OUTPUT_CHECK_TOL", None)
    assert level == 2 and residual > 0.0
    assert abs(s.log_determinant() - r.log_determinant()) <= 1e-12 * abs(r.log_determinant())
    z = rng.randn(N)
    assert abs(s.dot_solve(z) - r.dot_solve(z)) <= 1e-11 * abs(r.dot_solve(z))


@pytest.mark.parametrize("JR,JC,N,with_general", [(2, 16, 700, True), (1, 40, 900, False), (3, 60, 500, True), (2, 100, 400, True),
                                                  (0, 200, 300, False), (0, 512, 160, False)])
def test_row_distributed_factorisation_against_the_oracle_and_the_older_kernels(JR, JC, N, with_general):
    """Round 6: ``CholeskySolver.compute`` outside the chunked scans (general terms above width 32, every width above 64)
    keeps S in the REGISTERS of 1 / 4 / 16 / 64 workgroups that meet at one counter barrier per step (csrc/rows_kernels.hip;
    the reference's dynamic-width arm, cholesky.h:203).  The stored factor (phi, u, W, D), log det, ``dot_solve`` and
    ``solve`` against the oracle at widths 38 (general terms), 81, 127, 206 (general terms, four workgroups), 400 (sixteen)
    and 1024 (sixty-four) -- and against the one-workgroup kernels it replaces (S in LDS / in L2: ``CLR_NO_ROWS_KERNEL``)."""
    J = JR + 2 * JC
    case = synthetic(1, N, JR, JC, "accuracy", seed=J)
    cs = list(coeffs_of(case, 0))
    t, diag, y = case["t"][0], case["diag"][0] + 0.05, case["y"][0]
    np.random.seed(J)
    gen = NO_GENERAL
    if with_general:
        # four general rows that ARE a positive definite kernel: two undamped cosines a cos(w (t_i - t_j)) = U^T V with
        # U = a (cos w t, sin w t), V = (cos w t, sin w t), A = a.  (tests/test_celerite.py:102-105's polynomial rows with
        # random scalings are not: at N = 700 the recurrence amplifies rounding to 1e-5 of log det -- the oracle's own
        # double loops and a long-double run of them differ by that much, 4e-13 at N = 50.)
        a1, w1, a2, w2 = 0.3, 1.7, 0.2, 0.45
        U = np.vstack([a1 * np.cos(w1 * t), a1 * np.sin(w1 * t), a2 * np.cos(w2 * t), a2 * np.sin(w2 * t)])
        V = np.vstack([np.cos(w1 * t), np.sin(w1 * t), np.cos(w2 * t), np.sin(w2 * t)])
        gen = (np.full(N, a1 + a2), U, V)
    Jt = J + (4 if with_general else 0)
    r = ref.RefSolver()
    r.compute(0.1, *cs, *gen, t, diag)
    _, _, Jo, logdet, rphi, ru, rW, rD = r.state()
    assert Jo == Jt
    b = np.random.RandomState(J).randn(N, 2)
    want_solve, want_quad = r.solve(b), r.dot_solve(y)
    got = {}
    for rows in (True, False):
        batch.set_option("CLR_NO_ROWS_KERNEL", None if rows else "1")
        try:
            s = celerite_amd.CholeskySolver()
            s._hint_rhs(y)      # (GP.log_likelihood's way: the quadratic form of y out of the factorisation pass itself)
            s.compute(0.1, *cs, *gen, t, diag)
        finally:
            batch.set_option("CLR_NO_ROWS_KERNEL", None)
        tag = (Jt, "rows" if rows else "one workgroup")
        within("row-distributed factorisation: hinted dot_solve vs oracle", abs(s.dot_solve(y) - want_quad) / abs(want_quad), 1e-10, tag)
        want_b0 = r.dot_solve(b[:, 0])
        within("row-distributed factorisation: dot_solve vs oracle", abs(s.dot_solve(b[:, 0]) - want_b0) / abs(want_b0), 1e-10, tag)
        within("row-distributed factorisation: log det vs oracle", abs(s.log_determinant() - logdet) / abs(logdet), 1e-12, tag)
        within("row-distributed factorisation: dot_solve vs oracle", abs(s.dot_solve(y) - want_quad) / abs(want_quad), 1e-10, tag)
        within("row-distributed factorisation: solve vs oracle (of the largest entry)", np.max(np.abs(s.solve(b) - want_solve)) / np.max(np.abs(want_solve)), 1e-10, tag)
        st = s.__getstate__()
        phi, u = np.asarray(st[4]).reshape(rphi.shape), np.asarray(st[5]).reshape(ru.shape)
        W, D = np.asarray(st[6]).reshape(rW.shape), np.asarray(st[7])
        within("row-distributed factorisation: phi vs oracle", np.max(np.abs(phi - rphi)), 1e-15, tag)
        within("row-distributed factorisation: u vs oracle (of the largest entry)", np.max(np.abs(u - ru)) / np.max(np.abs(ru)), 1e-15, tag)
        within("row-distributed factorisation: W vs oracle (of the largest entry)", np.max(np.abs(W - rW)) / np.max(np.abs(rW)), 1e-10, tag)
        within("row-distributed factorisation: D vs oracle (relative)", np.max(np.abs(D - rD) / np.abs(rD)), 1e-10, tag)
        got[rows] = (s.log_determinant(), W, D)
    assert abs(got[True][0] - got[False][0]) <= 1e-12 * abs(logdet)
    # a pivot below zero is seen by every workgroup at the same step (cholesky.h:176)
    bad = list(cs)
    if JR:
        bad[0] = -20.0 * np.abs(bad[0])
    else:
        bad[2] = -20.0 * np.abs(bad[2])
    s3 = celerite_amd.CholeskySolver()
    with pytest.raises(LinAlgError):
        s3.compute(0.0, *bad, *gen, t, np.zeros(N))
    assert not s3.computed()


@pytest.mark.parametrize("JR,JC,N", [(0, 50, 3000), (2, 63, 2500), (1, 150, 2600)])
def test_diagonal_scans_with_several_waves_per_chunk(JR, JC, N):
    """Round 6: ``dot_L`` and ``dot`` on long series above width 64 -- the chunked diagonal scans with one thread per row,
    2 .. 16 waves per (chunk, right-hand side), the replay's sum over the rows through LDS (wsweep_kernels.hip); widths
    65 .. 128 walked the series sequentially before, above 128 the calls were refused.  cholesky.h:409-431, :533-560."""
    J = JR + 2 * JC
    case = synthetic(1, N, JR, JC, "accuracy", seed=J)
    cs = list(coeffs_of(case, 0))
    t, diag = case["t"][0], case["diag"][0] + 0.05
    r = ref.RefSolver()
    r.compute(0.1, *cs, *NO_GENERAL, t, diag)
    z = np.random.RandomState(J).randn(N, 3)
    s = celerite_amd.CholeskySolver()
    s.compute(0.1, *cs, *NO_GENERAL, t, diag)
    want = r.dot_L(z)
    within("diagonal scans, several waves: dot_L vs oracle (of the largest entry)", np.max(np.abs(s.dot_L(z) - want)) / np.max(np.abs(want)), 1e-11, J)
    want = r.dot(0.1, *cs, *NO_GENERAL, t, z)
    got = celerite_amd.CholeskySolver().dot(0.1, *cs, *NO_GENERAL, t, z)
    within("diagonal scans, several waves: dot vs oracle (of the largest entry)", np.max(np.abs(got - want)) / np.max(np.abs(want)), 1e-11, J)


@pytest.mark.parametrize("JR,JC,N", [(0, 40, 5000), (1, 63, 9000), (2, 100, 6000), (0, 200, 4500)])
def test_sweeps_above_width_64_as_affine_scans(JR, JC, N):
    """Round 6: ``dot_solve`` / ``solve`` (and ``predict``'s alpha) on long series above width 64 as chunked affine scans
    whose J x J chunk maps are built once per factor and direction (csrc/bigsweep_kernels.hip) -- against the oracle and
    against the sequential sweeps they replace (``CLR_NO_BIG_SWEEP``), several right-hand sides, a second call that
    reuses the maps, a new factor in the same solver.  cholesky.h:236-260, :343-357."""
    J = JR + 2 * JC
    case = synthetic(1, N, JR, JC, "accuracy", seed=J)
    cs = list(coeffs_of(case, 0))
    t, diag, y = case["t"][0], case["diag"][0] + 0.05, case["y"][0]
    rng = np.random.RandomState(J)
    b = rng.randn(N, 3)
    r = ref.RefSolver()
    r.compute(0.1, *cs, *NO_GENERAL, t, diag)
    want_solve, want_q = r.solve(b), [r.dot_solve(b[:, k]) for k in range(3)]
    s = celerite_amd.CholeskySolver()
    s.compute(0.1, *cs, *NO_GENERAL, t, diag)
    for attempt in range(2):      # (the second round reuses the chunk maps)
        for k in range(3):
            within("affine scans above width 64: dot_solve vs oracle", abs(s.dot_solve(b[:, k]) - want_q[k]) / abs(want_q[k]), 1e-11, (J, attempt))
        got = s.solve(b)
        within("affine scans above width 64: solve vs oracle (of the largest entry)", np.max(np.abs(got - want_solve)) / np.max(np.abs(want_solve)), 1e-11, (J, attempt))
    batch.set_option("CLR_NO_BIG_SWEEP", "1")
    try:
        seq = s.solve(b)
        seq_q = s.dot_solve(b[:, 0])
    finally:
        batch.set_option("CLR_NO_BIG_SWEEP", None)
    within("affine scans above width 64: solve vs the sequential sweep (of the largest entry)", np.max(np.abs(got - seq)) / np.max(np.abs(seq)), 1e-11, J)
    within("affine scans above width 64: dot_solve vs the sequential sweep", abs(s.dot_solve(b[:, 0]) - seq_q) / abs(seq_q), 1e-11, J)
    xs = np.sort(rng.uniform(t.min(), t.max(), 50))
    want_p = r.predict(y, xs)
    within("affine scans above width 64: predict vs oracle (of the largest entry)", np.max(np.abs(s.predict(y, xs) - want_p)) / np.max(np.abs(want_p)), 1e-11, J)
    # another factor in the same solver: the maps are rebuilt
    diag2 = diag * 1.3 + 0.01
    r.compute(0.05, *cs, *NO_GENERAL, t, diag2)
    s.compute(0.05, *cs, *NO_GENERAL, t, diag2)
    want2 = r.solve(b)
    within("affine scans above width 64: solve vs oracle (of the largest entry)", np.max(np.abs(s.solve(b) - want2)) / np.max(np.abs(want2)), 1e-11, (J, "new factor"))


def test_row_distributed_factorisations_from_several_host_threads():
    """The workgroups of one row-distributed factorisation above width 128 spin on each other and must all be resident;
    solvers of several host threads are serialised for that kernel (csrc/api_solver.hip): six threads, widths 300 / 520,
    every result against the oracle, no time-out status."""
    import threading
    cases = []
    for k, (JR, JC, N) in enumerate([(0, 150, 300), (2, 259, 200)] * 3):
        case = synthetic(1, N, JR, JC, "accuracy", seed=100 + k)
        cs = list(coeffs_of(case, 0))
        t, diag = case["t"][0], case["diag"][0] + 0.05
        r = ref.RefSolver()
        r.compute(0.1, *cs, *NO_GENERAL, t, diag)
        cases.append((cs, t, diag, r.log_determinant()))
    got, errors = [None] * len(cases), []

    def work(i):
        try:
            cs, t, diag, _ = cases[i]
            for _ in range(3):
                s = celerite_amd.CholeskySolver()
                s.compute(0.1, *cs, *NO_GENERAL, t, diag)
            got[i] = s.log_determinant()
        except Exception as e:  # (a lost workgroup would surface here as RuntimeError)
            errors.append(repr(e))

    threads = [threading.Thread(target=work, args=(i,)) for i in range(len(cases))]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors
    for i, (_, _, _, want) in enumerate(cases):
        within("row-distributed factorisation from several host threads: log det vs oracle", abs(got[i] - want) / abs(want), 1e-12, i)
