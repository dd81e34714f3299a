# -*- coding: utf-8 -*-
"""The Python producer of the hot path's inputs (celerite_amd.terms) against
(a) the coefficient vectors the REFERENCE's terms.py produces for the same
kernels (tests/golden/terms_golden.json, written by make_golden.py) and (b) the
reference's own assertions in tests/test_terms.py, re-authored."""
import json
import os
from itertools import product

import numpy as np
import pytest

from celerite_amd import terms

GOLDEN = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden",
                                     "terms_golden.json")))

T = terms


def build(key):
    return {
        "real": lambda: T.RealTerm(log_a=0.1, log_c=0.5),
        "real+real": lambda: T.RealTerm(log_a=0.1, log_c=0.5) + T.RealTerm(log_a=-0.1, log_c=0.7),
        "complex3": lambda: T.ComplexTerm(log_a=0.1, log_c=0.5, log_d=0.1),
        "complex4": lambda: T.ComplexTerm(log_a=0.1, log_b=-0.2, log_c=0.5, log_d=0.1),
        "jitter": lambda: T.JitterTerm(log_sigma=0.1),
        "sho_lowQ+jitter": lambda: T.SHOTerm(log_S0=0.1, log_Q=-1, log_omega0=0.5) + T.JitterTerm(log_sigma=0.1),
        "sho_lowQ": lambda: T.SHOTerm(log_S0=0.1, log_Q=-1, log_omega0=0.5),
        "sho_highQ": lambda: T.SHOTerm(log_S0=0.1, log_Q=1.0, log_omega0=0.5),
        "sho+real": lambda: T.SHOTerm(log_S0=0.1, log_Q=1.0, log_omega0=0.5) + T.RealTerm(log_a=0.1, log_c=0.4),
        "sho*real": lambda: T.SHOTerm(log_S0=0.1, log_Q=1.0, log_omega0=0.5) * T.RealTerm(log_a=0.1, log_c=0.4),
        "real+sho (config 1)": lambda: T.RealTerm(0.1, 0.5) + T.SHOTerm(log_S0=0.1, log_Q=1.0, log_omega0=0.5),
        "matern32": lambda: T.Matern32Term(log_sigma=0.3, log_rho=-0.4),
        "matern32_eps": lambda: T.Matern32Term(log_sigma=0.3, log_rho=-0.4, eps=0.1),
        "complex*complex": lambda: T.ComplexTerm(0.2, -3.0, 0.5, 0.01) * T.ComplexTerm(0.6, 0.7, 1.0),
        "(real+complex)*sho": lambda: (T.RealTerm(log_a=0.1, log_c=0.5) + T.ComplexTerm(0.2, -3.0, 0.5, 0.01)) * T.SHOTerm(1.0, 0.2, 3.0),
        "bench width 8": lambda: (T.RealTerm(1.0, 0.1) + T.RealTerm(1.0, 0.1) + T.ComplexTerm(0.1, 2.0, 1.6)
                                  + T.ComplexTerm(0.1, 2.0, 1.6) + T.ComplexTerm(0.1, 2.0, 1.6)),
        "test_log_likelihood full kernel (width 26)": full_kernel,
    }[key]()


def full_kernel():
    kernel = T.RealTerm(0.1, 0.5)
    termlist = [(0.1 + 10. / j, 0.5 + 10. / j) for j in range(1, 4)]
    termlist += [(1.0 + 10. / j, 0.01 + 10. / j, 0.5, 0.01) for j in range(1, 10)]
    termlist += [(0.6, 0.7, 1.0), (0.3, 0.05, 0.5, 0.6)]
    for term in termlist:
        kernel += T.ComplexTerm(*term) if len(term) > 2 else T.RealTerm(*term)
    return kernel


@pytest.mark.parametrize("key", sorted(GOLDEN))
def test_coefficients_match_reference(key):
    want = GOLDEN[key]
    k = build(key)
    assert list(k.get_parameter_names(include_frozen=True)) == want["parameter_names"]
    assert np.array_equal(k.get_parameter_vector(include_frozen=True), want["parameter_vector"])
    got = k.coefficients
    assert len(got) == 6
    for g, w in zip(got, want["coefficients"]):
        assert len(g) == len(w)
        # same formulas, same libm: bit-for-bit
        assert np.array_equal(np.asarray(g, dtype=float), np.asarray(w, dtype=float))
    assert k.jitter == want["jitter"]
    assert repr(k) == want["repr"]


def test_product():  # tests/test_terms.py:13-34
    np.random.seed(42)
    t = np.sort(np.random.uniform(0, 5, 100))
    tau = t[:, None] - t[None, :]
    k1 = terms.RealTerm(log_a=0.1, log_c=0.5)
    k2 = terms.ComplexTerm(0.2, -3.0, 0.5, 0.01)
    k3 = terms.SHOTerm(1.0, 0.2, 3.0)
    K1, K2, K3 = k1.get_value(tau), k2.get_value(tau), k3.get_value(tau)
    assert np.allclose((k1 + k2).get_value(tau), K1 + K2)
    assert np.allclose((k3 + k2).get_value(tau), K3 + K2)
    assert np.allclose((k1 + k2 + k3).get_value(tau), K1 + K2 + K3)
    for (a, b), (A, B) in zip(product((k1, k2, k3, k1 + k2, k1 + k3, k2 + k3), (k1, k2, k3)),
                              product((K1, K2, K3, K1 + K2, K1 + K3, K2 + K3), (K1, K2, K3))):
        assert np.allclose((a * b).get_value(tau), A * B)


def test_bounds():  # tests/test_terms.py:37-46
    bounds = [(-1.0, 0.3), (-2.0, 5.0)]
    kernel = terms.RealTerm(log_a=0.1, log_c=0.5, bounds=bounds)
    b0 = kernel.get_parameter_bounds()
    assert all(np.allclose(a, b) for a, b in zip(b0, bounds))
    kernel = terms.RealTerm(log_a=0.1, log_c=0.5, bounds=dict(zip(["log_a", "log_c"], bounds)))
    assert all(np.allclose(a, b) for a, b in zip(b0, kernel.get_parameter_bounds()))
    with pytest.raises(ValueError):
        terms.RealTerm(log_a=0.5, log_c=0.5, bounds=bounds)  # outside => non-finite prior


JAC_KERNELS = [  # tests/test_terms.py:49-63 (+ a Matern-3/2 product, a frozen parameter and a nested product)
    lambda: terms.RealTerm(log_a=0.1, log_c=0.5),
    lambda: terms.RealTerm(log_a=0.1, log_c=0.5) + terms.RealTerm(log_a=-0.1, log_c=0.7),
    lambda: terms.ComplexTerm(log_a=0.1, log_c=0.5, log_d=0.1),
    lambda: terms.ComplexTerm(log_a=0.1, log_b=-0.2, log_c=0.5, log_d=0.1),
    lambda: terms.SHOTerm(log_S0=0.1, log_Q=-1, log_omega0=0.5),
    lambda: terms.SHOTerm(log_S0=0.1, log_Q=1.0, log_omega0=0.5),
    lambda: terms.SHOTerm(log_S0=0.1, log_Q=1.0, log_omega0=0.5) + terms.RealTerm(log_a=0.1, log_c=0.4),
    lambda: terms.SHOTerm(log_S0=0.1, log_Q=1.0, log_omega0=0.5) * terms.RealTerm(log_a=0.1, log_c=0.4),
    lambda: terms.Matern32Term(log_sigma=0.3, log_rho=-0.2) * terms.SHOTerm(log_S0=0.1, log_Q=1.0, log_omega0=0.5)
    + terms.JitterTerm(log_sigma=0.1),
    lambda: (terms.SHOTerm(log_S0=0.1, log_Q=-1, log_omega0=0.5) + terms.RealTerm(log_a=0.1, log_c=0.4))
    * terms.ComplexTerm(log_a=0.1, log_b=-0.2, log_c=0.5, log_d=0.1),
]


def _central(fn, k, eps=1.34e-7):
    """Richardson-extrapolated central differences of fn() with respect to k's parameter vector (h and h / 2)."""
    v = k.get_parameter_vector()
    rows = []
    for i, pval in enumerate(v):
        d = []
        for h in (eps, 0.5 * eps):
            v[i] = pval + h
            k.set_parameter_vector(v)
            hi = fn()
            v[i] = pval - h
            k.set_parameter_vector(v)
            d.append((hi - fn()) / (2.0 * h))
            v[i] = pval
        rows.append((4.0 * d[1] - d[0]) / 3.0)
    k.set_parameter_vector(v)
    return np.array(rows)


@pytest.mark.parametrize("make", JAC_KERNELS)
def test_jacobian(make):  # tests/test_terms.py:64-84 -- here WITHOUT autograd: the built-in formulas on dual numbers
    k = make()
    v = k.get_parameter_vector()
    c = np.concatenate(k.coefficients)
    jac = k.get_coeffs_jacobian()
    assert jac.shape == (len(v), len(c))
    jac0 = _central(lambda: np.concatenate(k.coefficients), k)
    assert np.allclose(jac, jac0, rtol=1e-6, atol=1e-7 * max(1.0, np.max(np.abs(c))))
    # a frozen parameter drops its row (terms.py:214-215); include_frozen keeps it
    name = k.get_parameter_names()[0]
    k.freeze_parameter(name)
    assert np.array_equal(k.get_coeffs_jacobian(), jac[1:]) and np.array_equal(k.get_coeffs_jacobian(include_frozen=True), jac)


@pytest.mark.parametrize("make", [  # tests/test_terms.py:87-94
    lambda: terms.JitterTerm(log_sigma=0.5),
    lambda: terms.RealTerm(log_a=0.5, log_c=0.1),
    lambda: terms.RealTerm(log_a=0.5, log_c=0.1) + terms.JitterTerm(log_sigma=0.3),
    lambda: terms.JitterTerm(log_sigma=0.5) + terms.JitterTerm(log_sigma=0.1),
])
def test_jitter_jacobian(make):  # tests/test_terms.py:95-119
    k = make()
    jac = k.get_jitter_jacobian()
    assert len(jac) == len(k.get_parameter_vector())
    assert np.allclose(jac, _central(lambda: k.jitter, k), rtol=1e-7, atol=1e-9)


def test_jacobian_matches_the_golden_coefficients():
    """The dual-number evaluation returns the very coefficients tests/golden/terms_golden.json pins (generated from the
    reference's terms.py): value parts bit-identical to get_all_coefficients, whose parity test_reference_coefficients
    asserts -- so the Jacobian differentiates the golden-pinned function, not a restatement of it."""
    for make in JAC_KERNELS:
        k = make()
        res, n = k._dual_all()
        assert res is not None
        vals = np.array([x.v for blk in res[0] for x in blk])
        assert np.allclose(vals, np.concatenate(k.coefficients), rtol=1e-15, atol=0.0)
        assert np.isclose(terms._Dual.lift(res[1], n).v, k.jitter, rtol=1e-15, atol=0.0)


def test_user_defined_terms_still_need_autograd():  # terms.py:197-215: the reference's behaviour for everything else
    class Custom(terms.Term):
        parameter_names = ("log_a", )

        def get_real_coefficients(self, params):
            return np.exp(params[0]), 1.0

    class Tweaked(terms.RealTerm):
        def get_real_coefficients(self, params):
            return 2.0 * np.exp(params[0]), np.exp(params[1])

    if terms.HAS_AUTOGRAD:
        pytest.skip("autograd present")
    for k in (Custom(log_a=0.1), Tweaked(log_a=0.1, log_c=0.2), Custom(log_a=0.1) + terms.RealTerm(log_a=0.1, log_c=0.5)):
        with pytest.raises(ImportError):
            k.get_coeffs_jacobian()
        with pytest.raises(ImportError):
            k.get_jitter_jacobian()


def test_quiet():  # tests/test_terms.py:122-139
    terms.RealTerm(log_a=0.1, log_c=0.5, quiet=True)
    terms.RealTerm(0.1, 0.5, quiet=True)
    with pytest.raises(ValueError):
        terms.ComplexTerm(log_a=1.0, log_b=10.0, log_c=1.0, log_d=1.0)
    with pytest.raises(ValueError):
        terms.ComplexTerm(log_a=1.0, log_b=10.0, log_c=1.0, log_d=1.0, quiet=False)
    terms.ComplexTerm(log_a=1.0, log_b=10.0, log_c=1.0, log_d=1.0, quiet=True)


def test_jitter_products_rejected():
    with pytest.raises(ValueError):
        terms.JitterTerm(log_sigma=0.1) * terms.RealTerm(0.1, 0.5)


def test_psd_and_sturm_check():
    k = terms.SHOTerm(log_S0=0.1, log_Q=1.0, log_omega0=0.5)
    S0, Q, w0 = np.exp(0.1), np.exp(1.0), np.exp(0.5)
    w = np.array([0.1, 0.7, 1.6, 3.0])
    want = np.sqrt(2 / np.pi) * S0 * w0 ** 4 / ((w ** 2 - w0 ** 2) ** 2 + w0 ** 2 * w ** 2 / Q ** 2)
    assert np.allclose(k.get_psd(w), want, rtol=1e-12)
    assert k.check_parameters()
    # a complex term with a c < b d has negative power somewhere
    bad = terms.ComplexTerm(log_a=0.0, log_b=3.0, log_c=0.0, log_d=0.0, quiet=True)
    assert not bad.check_parameters()
    # two reals with one negative amplitude can still be valid (SHO, Q < 1/2)
    assert terms.SHOTerm(log_S0=0.1, log_Q=-1, log_omega0=0.5).check_parameters()
