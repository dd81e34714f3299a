# -*- coding: utf-8 -*-
"""celerite kernel terms: log-parameters -> the six coefficient blocks.

Own implementation of the interface of the reference's ``celerite/terms.py``:
every term maps its parameter vector to
``(a_real, c_real, a_comp, b_comp, c_comp, d_comp)`` (+ a scalar jitter) such
that ``k(tau) = sum_j a_j exp(-c_j tau)
+ sum_k exp(-c_k tau) [a_k cos(d_k tau) + b_k sin(d_k tau)]``.
Those arrays are the *inputs* of the hot path (``CholeskySolver.compute``,
``celerite/celerite.py:133-157``); tests/golden/terms_golden.json holds the
values the reference produces for the same kernels and tests/test_terms.py
checks this module against them.

Formulas (reference file:line): RealTerm ``terms.py:389-391``; ComplexTerm
``:435-444``; SHOTerm ``:489-517``; Matern32Term ``:562-566``; JitterTerm
``:356-357``; sum = concatenation ``:304-313``; product algebra ``:234-277``.
"""
try:  # the reference switches to autograd.numpy when it is importable (terms.py:5-13)
    import autograd  # noqa: F401
except ImportError:
    import numpy as np
    HAS_AUTOGRAD = False
else:  # pragma: no cover - autograd is not part of this image
    import autograd.numpy as np
    from autograd import jacobian, elementwise_grad
    HAS_AUTOGRAD = True

from .modeling import Model, ModelSet
from .solver import get_kernel_value, get_psd_value, check_coefficients

__all__ = [
    "Term", "TermProduct", "TermSum",
    "JitterTerm", "RealTerm", "ComplexTerm", "SHOTerm", "Matern32Term",
]

_EMPTY2 = lambda: (np.empty(0), np.empty(0))
_EMPTY4 = lambda: (np.empty(0), np.empty(0), np.empty(0), np.empty(0))


# ---- forward-mode derivatives of the coefficient formulas -----------------------------------------------------
# The reference obtains d(coefficients) / d(parameters) from autograd (terms.py:197-215) and raises ImportError
# without it.  autograd is not part of the MI355X image, and the device gradient (clr_batch_grad /
# CholeskySolver.grad_log_likelihood) is with respect to the COEFFICIENTS: the chain rule to the kernel's
# log-parameters is closed here, by evaluating the built-in terms' own formulas on dual numbers (value + gradient
# with respect to the full parameter vector) -- exact to rounding, no finite differences.  User-defined terms still
# go through autograd, exactly as in the reference.
import numpy as _np


class _Dual(object):
    """value + gradient (1-D array over the full parameter vector); just the arithmetic the term formulas use"""
    __slots__ = ("v", "g")
    __array_priority__ = 1000.0

    def __init__(self, v, g):
        self.v = float(v)
        self.g = g

    @staticmethod
    def lift(x, n):
        return x if isinstance(x, _Dual) else _Dual(x, _np.zeros(n))

    def _o(self, other):
        return other if isinstance(other, _Dual) else _Dual(other, _np.zeros_like(self.g))

    def __add__(self, o):
        o = self._o(o)
        return _Dual(self.v + o.v, self.g + o.g)
    __radd__ = __add__

    def __neg__(self):
        return _Dual(-self.v, -self.g)

    def __sub__(self, o):
        o = self._o(o)
        return _Dual(self.v - o.v, self.g - o.g)

    def __rsub__(self, o):
        return self._o(o) - self

    def __mul__(self, o):
        o = self._o(o)
        return _Dual(self.v * o.v, self.v * o.g + o.v * self.g)
    __rmul__ = __mul__

    def __truediv__(self, o):
        o = self._o(o)
        return _Dual(self.v / o.v, (self.g - (self.v / o.v) * o.g) / o.v)

    def __rtruediv__(self, o):
        return self._o(o) / self


def _dexp(x):
    e = _np.exp(x.v)
    return _Dual(e, e * x.g)


def _dsqrt(x):
    r = _np.sqrt(x.v)
    return _Dual(r, x.g / (2.0 * r))


def _dual_params(values):
    n = len(values)
    eye = _np.eye(n)
    return [_Dual(values[i], eye[i]) for i in range(n)]


def _embed(blocks, offset, total):
    """gradients of a sub-term's coefficients (w.r.t. its own parameters) inside the parent's parameter vector"""
    out = []
    for blk in blocks:
        row = []
        for x in blk:
            g = _np.zeros(total)
            g[offset:offset + len(x.g)] = x.g
            row.append(_Dual(x.v, g))
        out.append(row)
    return out


class Term(Model):
    """Base class: an empty kernel.  Subclasses override
    :meth:`get_real_coefficients` and/or :meth:`get_complex_coefficients`."""

    _has_jitter = False
    _has_coeffs = True

    @property
    def terms(self):
        return [self]

    # -- evaluation (host-side scalar helpers; cpp/include/celerite/utils.h:106-163) --
    def get_value(self, tau):
        tau = np.asarray(tau)
        k = get_kernel_value(*(list(self.coefficients) + [tau.flatten()]))
        return np.asarray(k).reshape(tau.shape)

    def get_psd(self, omega):
        w = np.asarray(omega)
        p = get_psd_value(*(list(self.coefficients) + [w.flatten()]))
        return p.reshape(w.shape)

    def check_parameters(self):
        """Sturm-sequence test for a non-negative PSD (utils.h:27-104)."""
        return check_coefficients(*self.coefficients)

    # -- algebra ---------------------------------------------------------------------
    def __add__(self, other):
        return TermSum(self, other)

    def __radd__(self, other):
        return TermSum(other, self)

    def __mul__(self, other):
        return TermProduct(self, other)

    def __rmul__(self, other):
        return TermProduct(other, self)

    # -- coefficient blocks ---------------------------------------------------------------
    def get_real_coefficients(self, params):
        return _EMPTY2()

    def get_complex_coefficients(self, params):
        return _EMPTY4()

    def get_all_coefficients(self, params=None):
        if params is None:
            params = self.get_parameter_vector(include_frozen=True)
        real = self.get_real_coefficients(params)
        comp = self.get_complex_coefficients(params)
        if len(comp) == 3:  # b omitted => b = 0
            a, c, d = comp
            comp = (a, np.zeros_like(a), c, d)
        return [np.atleast_1d(block) for block in tuple(real) + tuple(comp)]

    @property
    def coefficients(self):
        """The six 1-D blocks, validated."""
        blocks = self.get_all_coefficients(self.get_parameter_vector(include_frozen=True))
        if len(blocks) != 6:
            raise ValueError("there must be 6 coefficient blocks")
        if any(b.ndim != 1 for b in blocks):
            raise ValueError("coefficient blocks must be 1D")
        if len(blocks[0]) != len(blocks[1]):
            raise ValueError("coefficient blocks must have the same shape")
        if any(len(b) != len(blocks[2]) for b in blocks[3:]):
            raise ValueError("coefficient blocks must have the same shape")
        return blocks

    def get_jitter(self, params):
        return 0.0

    @property
    def jitter(self):
        return self.get_jitter(self.get_parameter_vector(include_frozen=True))

    # -- Jacobians (terms.py:197-215) --------------------------------------------------------
    # Built-in terms (and sums / products of them): their formulas on dual numbers, no dependency.  Anything else:
    # autograd, exactly like the reference -- ImportError without it.
    def _dual_coefficients(self, p):
        """The six coefficient blocks as lists of ``_Dual`` and the jitter as a ``_Dual``, for the parameter duals
        ``p`` -- or None when this term's formulas are not known here (user-defined terms)."""
        return None

    def _dual_all(self):
        vec = _np.asarray(self.get_parameter_vector(include_frozen=True), dtype=float)
        return self._dual_coefficients(_dual_params(vec)), len(vec)

    def get_jitter_jacobian(self, include_frozen=False):
        res, n = self._dual_all()
        if res is not None:
            jac = _np.array(_Dual.lift(res[1], n).g, dtype=float)
        else:
            if not HAS_AUTOGRAD:
                raise ImportError("'autograd' must be installed to compute gradients")
            jac = elementwise_grad(self.get_jitter)(self.get_parameter_vector(include_frozen=True))
        return jac if include_frozen else jac[self.unfrozen_mask]

    def get_coeffs_jacobian(self, include_frozen=False):
        res, n = self._dual_all()
        if res is not None:
            flat = [x for blk in res[0] for x in blk]
            jac = _np.array([x.g for x in flat], dtype=float).reshape(len(flat), n).T
        else:
            if not HAS_AUTOGRAD:
                raise ImportError("'autograd' must be installed to compute gradients")
            flat = lambda p: np.concatenate(self.get_all_coefficients(p))
            jac = jacobian(flat)(self.get_parameter_vector(include_frozen=True)).T
        return jac if include_frozen else jac[self.unfrozen_mask]

    def _formulas_are(self, cls):
        """True when this object's coefficient formulas are the ones `cls` defines (not overridden by a subclass)."""
        t = type(self)
        return all(getattr(t, m) is getattr(cls, m)
                   for m in ("get_real_coefficients", "get_complex_coefficients", "get_all_coefficients", "get_jitter"))


class TermSum(Term, ModelSet):
    """k1 + k2 + ...: the coefficient blocks are concatenated term by term."""

    def __init__(self, *terms):
        flat = []
        for term in terms:
            flat.extend(term.terms)
        ModelSet.__init__(self, [("terms[{0}]".format(i), t) for i, t in enumerate(flat)])

    def __repr__(self):
        return "(" + " + ".join("{0}".format(t) for t in self.terms) + ")"

    @property
    def terms(self):
        return list(self.models.values())

    @property
    def _has_jitter(self):
        return any(t._has_jitter for t in self.models.values())

    @property
    def _has_coeffs(self):
        return any(t._has_coeffs for t in self.models.values())

    def _split(self, params):
        start = 0
        for term in self.models.values():
            stop = start + term.full_size
            yield term, params[start:stop]
            start = stop

    def get_all_coefficients(self, params=None):
        if params is None:
            params = self.get_parameter_vector(include_frozen=True)
        per_term = [term.get_all_coefficients(p) for term, p in self._split(params)]
        return [np.concatenate(blocks) for blocks in zip(*per_term)]

    def get_jitter(self, params=None):
        if params is None:
            params = self.get_parameter_vector(include_frozen=True)
        total = 0.0
        for term, p in self._split(params):
            total += term.get_jitter(p)
        return total

    def _dual_coefficients(self, p):
        if not self._formulas_are(TermSum):
            return None
        n = len(p)
        blocks, jitter, start = [[] for _ in range(6)], _Dual(0.0, _np.zeros(n)), 0
        for term in self.models.values():
            m = term.full_size
            sub = term._dual_coefficients(_dual_params([x.v for x in p[start:start + m]]))
            if sub is None:
                return None
            for dst, src in zip(blocks, _embed(sub[0], start, n)):
                dst.extend(src)
            jitter = jitter + _embed([[_Dual.lift(sub[1], m)]], start, n)[0][0]
            start += m
        return blocks, jitter


class TermProduct(Term, ModelSet):
    """k1 * k2, expanded back into real + complex celerite terms.

    real x real         a1 a2 e^{-(c1+c2) tau}
    real x complex      (a1 a2, a1 b2, c1 + c2, d2)
    complex x complex   product-to-sum: two complex terms at d1 -+ d2.
    """

    def __init__(self, k1, k2):
        if k1._has_jitter or k2._has_jitter:
            raise ValueError("Products are not implemented for terms with jitter")
        ModelSet.__init__(self, [("k1", k1), ("k2", k2)])

    def __repr__(self):
        return "{0} * {1}".format(self.models["k1"], self.models["k2"])

    @property
    def terms(self):
        return [self]

    def get_all_coefficients(self, params=None):
        if params is None:
            params = self.get_parameter_vector(include_frozen=True)
        k1, k2 = self.models["k1"], self.models["k2"]
        n1 = k1.full_size
        ar1, cr1, ac1, bc1, cc1, dc1 = k1.get_all_coefficients(params[:n1])
        ar2, cr2, ac2, bc2, cc2, dc2 = k2.get_all_coefficients(params[n1:])
        reals1, reals2 = list(zip(ar1, cr1)), list(zip(ar2, cr2))
        comps1, comps2 = list(zip(ac1, bc1, cc1, dc1)), list(zip(ac2, bc2, cc2, dc2))

        ar, cr, ac, bc, cc, dc = [], [], [], [], [], []
        for a1, c1 in reals1:
            for a2, c2 in reals2:
                ar.append(a1 * a2)
                cr.append(c1 + c2)

        # every real factor of one side against every complex factor of the other
        for rs, cs in ((reals1, comps2), (reals2, comps1)):
            for a1, c1 in rs:
                for a2, b2, c2, d2 in cs:
                    ac.append(a1 * a2)
                    bc.append(a1 * b2)
                    cc.append(c1 + c2)
                    dc.append(d2)

        for a1, b1, c1, d1 in comps1:
            for a2, b2, c2, d2 in comps2:
                for sign in (-1.0, 1.0):
                    ac.append(0.5 * (a1 * a2 - sign * b1 * b2))
                    bc.append(0.5 * (b1 * a2 + sign * a1 * b2))
                    cc.append(c1 + c2)
                    dc.append(d1 + sign * d2)

        return [np.array(block) for block in (ar, cr, ac, bc, cc, dc)]

    def _dual_coefficients(self, p):
        if not self._formulas_are(TermProduct):
            return None
        k1, k2 = self.models["k1"], self.models["k2"]
        n, n1 = len(p), k1.full_size
        s1 = k1._dual_coefficients(_dual_params([x.v for x in p[:n1]]))
        s2 = k2._dual_coefficients(_dual_params([x.v for x in p[n1:]]))
        if s1 is None or s2 is None:
            return None
        ar1, cr1, ac1, bc1, cc1, dc1 = _embed(s1[0], 0, n)
        ar2, cr2, ac2, bc2, cc2, dc2 = _embed(s2[0], n1, n)
        reals1, reals2 = list(zip(ar1, cr1)), list(zip(ar2, cr2))
        comps1, comps2 = list(zip(ac1, bc1, cc1, dc1)), list(zip(ac2, bc2, cc2, dc2))
        ar, cr, ac, bc, cc, dc = [], [], [], [], [], []
        for a1, c1 in reals1:                       # (the algebra of get_all_coefficients, on duals)
            for a2, c2 in reals2:
                ar.append(a1 * a2)
                cr.append(c1 + c2)
        for rs, cs in ((reals1, comps2), (reals2, comps1)):
            for a1, c1 in rs:
                for a2, b2, c2, d2 in cs:
                    ac.append(a1 * a2)
                    bc.append(a1 * b2)
                    cc.append(c1 + c2)
                    dc.append(d2)
        for a1, b1, c1, d1 in comps1:
            for a2, b2, c2, d2 in comps2:
                for sign in (-1.0, 1.0):
                    ac.append(0.5 * (a1 * a2 - sign * (b1 * b2)))
                    bc.append(0.5 * (b1 * a2 + sign * (a1 * b2)))
                    cc.append(c1 + c2)
                    dc.append(d1 + sign * d2)
        return [ar, cr, ac, bc, cc, dc], _Dual(0.0, _np.zeros(n))


class JitterTerm(Term):
    """White noise ``sigma^2 delta_nm``; parameter ``log_sigma``."""

    _has_jitter = True
    _has_coeffs = False
    parameter_names = ("log_sigma", )

    def __repr__(self):
        return "JitterTerm({0.log_sigma})".format(self)

    def get_jitter(self, params):
        return np.exp(2.0 * params[0])

    def _dual_coefficients(self, p):
        if not self._formulas_are(JitterTerm):
            return None
        return [[] for _ in range(6)], _dexp(2.0 * p[0])


class RealTerm(Term):
    """``a exp(-c tau)``; parameters ``log_a``, ``log_c``."""

    parameter_names = ("log_a", "log_c")

    def __repr__(self):
        return "RealTerm({0.log_a}, {0.log_c})".format(self)

    def get_real_coefficients(self, params):
        return np.exp(params[0]), np.exp(params[1])

    def _dual_coefficients(self, p):
        if not self._formulas_are(RealTerm):
            return None
        return [[_dexp(p[0])], [_dexp(p[1])], [], [], [], []], 0.0


class ComplexTerm(Term):
    """``exp(-c tau) [a cos(d tau) + b sin(d tau)]``.

    Parameters ``log_a, log_b, log_c, log_d``; with three arguments (or no
    ``log_b`` keyword) ``b = 0`` and is not a parameter.  On its own the term is
    positive definite only if ``a c >= b d``: :meth:`log_prior` enforces it.
    """

    def __init__(self, *args, **kwargs):
        self.fit_b = len(args) == 4 or "log_b" in kwargs
        if self.fit_b:
            self.parameter_names = ("log_a", "log_b", "log_c", "log_d")
        else:
            self.parameter_names = ("log_a", "log_c", "log_d")
        super(ComplexTerm, self).__init__(*args, **kwargs)

    def __repr__(self):
        if self.fit_b:
            return "ComplexTerm({0.log_a}, {0.log_b}, {0.log_c}, {0.log_d})".format(self)
        return "ComplexTerm({0.log_a}, {0.log_c}, {0.log_d})".format(self)

    def get_complex_coefficients(self, params):
        if self.fit_b:
            log_a, log_b, log_c, log_d = params
            return np.exp(log_a), np.exp(log_b), np.exp(log_c), np.exp(log_d)
        log_a, log_c, log_d = params
        return np.exp(log_a), 0.0, np.exp(log_c), np.exp(log_d)

    def _dual_coefficients(self, p):
        if not self._formulas_are(ComplexTerm):
            return None
        n = len(p)
        if self.fit_b:
            return [[], [], [_dexp(p[0])], [_dexp(p[1])], [_dexp(p[2])], [_dexp(p[3])]], 0.0
        return [[], [], [_dexp(p[0])], [_Dual(0.0, _np.zeros(n))], [_dexp(p[1])], [_dexp(p[2])]], 0.0

    def log_prior(self):
        if self.fit_b and self.log_a + self.log_c < self.log_b + self.log_d:
            return -np.inf
        return super(ComplexTerm, self).log_prior()


class SHOTerm(Term):
    """Stochastically driven damped harmonic oscillator,
    ``S(w) = sqrt(2/pi) S0 w0^4 / ((w^2 - w0^2)^2 + w0^2 w^2 / Q^2)``.

    Parameters ``log_S0, log_Q, log_omega0``.  Over-damped (Q < 1/2): two real
    terms; otherwise one complex term.
    """

    parameter_names = ("log_S0", "log_Q", "log_omega0")

    def __repr__(self):
        return "SHOTerm({0.log_S0}, {0.log_Q}, {0.log_omega0})".format(self)

    def get_real_coefficients(self, params):
        log_S0, log_Q, log_omega0 = params
        Q = np.exp(log_Q)
        if Q >= 0.5:
            return _EMPTY2()
        S0, w0 = np.exp(log_S0), np.exp(log_omega0)
        f = np.sqrt(1.0 - 4.0 * Q**2)
        return (0.5 * S0 * w0 * Q * np.array([1.0 + 1.0 / f, 1.0 - 1.0 / f]),
                0.5 * w0 / Q * np.array([1.0 - f, 1.0 + f]))

    def get_complex_coefficients(self, params):
        log_S0, log_Q, log_omega0 = params
        Q = np.exp(log_Q)
        if Q < 0.5:
            return _EMPTY4()
        S0, w0 = np.exp(log_S0), np.exp(log_omega0)
        f = np.sqrt(4.0 * Q**2 - 1)
        return (S0 * w0 * Q, S0 * w0 * Q / f, 0.5 * w0 / Q, 0.5 * w0 / Q * f)

    def _dual_coefficients(self, p):
        if not self._formulas_are(SHOTerm):
            return None
        S0, Q, w0 = _dexp(p[0]), _dexp(p[1]), _dexp(p[2])
        if Q.v < 0.5:       # (the formulas of get_real_coefficients)
            f = _dsqrt(1.0 - 4.0 * (Q * Q))
            pre, rate = 0.5 * (S0 * w0 * Q), 0.5 * (w0 / Q)
            return [[pre * (1.0 + 1.0 / f), pre * (1.0 - 1.0 / f)], [rate * (1.0 - f), rate * (1.0 + f)],
                    [], [], [], []], 0.0
        f = _dsqrt(4.0 * (Q * Q) - 1.0)   # (get_complex_coefficients)
        a = S0 * w0 * Q
        return [[], [], [a], [a / f], [0.5 * (w0 / Q)], [0.5 * (w0 / Q) * f]], 0.0


class Matern32Term(Term):
    """Approximate Matern-3/2: a complex term with ``d = eps`` -> 0.

    Parameters ``log_sigma``, ``log_rho``; ``eps`` (default 0.01) is fixed.
    """

    parameter_names = ("log_sigma", "log_rho")

    def __init__(self, *args, **kwargs):
        eps = kwargs.pop("eps", 0.01)
        super(Matern32Term, self).__init__(*args, **kwargs)
        self.eps = eps

    def __repr__(self):
        return "Matern32Term({0.log_sigma}, {0.log_rho}, eps={0.eps})".format(self)

    def get_complex_coefficients(self, params):
        log_sigma, log_rho = params
        w0 = np.sqrt(3.0) * np.exp(-log_rho)
        S0 = np.exp(2.0 * log_sigma) / w0
        return (w0 * S0, w0 * w0 * S0 / self.eps, w0, self.eps)

    def _dual_coefficients(self, p):
        if not self._formulas_are(Matern32Term):
            return None
        w0 = _np.sqrt(3.0) * _dexp(-p[1])
        S0 = _dexp(2.0 * p[0]) / w0
        return [[], [], [w0 * S0], [w0 * w0 * S0 / self.eps], [w0], [_Dual(self.eps, _np.zeros(len(p)))]], 0.0
