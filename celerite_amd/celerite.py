# -*- coding: utf-8 -*-
"""The ``GP`` front-end: the Python caller of the hot path.

Own implementation of the interface of the reference's
``celerite/celerite.py`` (``class GP`` :14-567).  What matters for the hot path
 is reproduced exactly:

* the argument list handed to ``solver.compute`` (celerite.py:144-157):
  ``(jitter, a_real, c_real, a_comp, b_comp, c_comp, d_comp, A, U, V, t,
  yerr**2)`` with ``np.empty(0)`` / ``np.empty((0, 0))`` for absent general
  terms;
* the errors: ``ValueError("the input coordinates must be sorted")``,
  ``ValueError("dimension mismatch")``, ``RuntimeError("you must call
  'compute' first")``, ``solver.LinAlgError`` (or ``-inf`` when ``quiet``);
* ``log_likelihood = -0.5 (r^T K^-1 r + log det K + N log 2 pi)`` with
  non-finite results mapped to ``-inf`` (celerite.py:211-218);
* the ``dirty`` / ``computed`` life-cycle (celerite.py:86-101,160-171).

``solver`` is ``celerite_amd.solver`` -- the compiled module whose
``CholeskySolver`` runs the factorisation and the solves as HIP kernels on an
MI355X through the C ABI of ``include/celerite_hip.h``.  For many likelihood
evaluations (the data-parallel axis) use :mod:`celerite_amd.batch`, which
bypasses per-problem Python entirely.
"""
import math
import warnings

import numpy as np

from . import solver, terms
from .modeling import ModelSet, ConstantModel

__all__ = ["GP"]

_LOG_2PI = math.log(2.0 * math.pi)


def _no_general():
    return np.empty(0), np.empty((0, 0)), np.empty((0, 0))


class GP(ModelSet):
    """A celerite Gaussian process: ``kernel`` (a :class:`terms.Term`) + ``mean``.

    Args:
        kernel: the covariance model.
        mean: a float or a :class:`modeling.Model` (default ``0.0``).
        fit_mean: if ``False`` (default) every parameter of ``mean`` is frozen.
        log_white_noise / fit_white_noise: deprecated spelling of adding a
            :class:`terms.JitterTerm`.
    """

    def __init__(self, kernel, mean=0.0, fit_mean=False,
                 log_white_noise=None, fit_white_noise=False):
        self._solver = None
        self._computed = False
        self._t = None
        self._y_var = None

        if log_white_noise is not None:
            warnings.warn("The 'log_white_noise' parameter is deprecated. "
                          "Use a 'JitterTerm' instead.")
            noise = terms.JitterTerm(log_sigma=float(log_white_noise))
            if not fit_white_noise:
                noise.freeze_parameter("log_sigma")
            kernel += noise

        try:
            constant = float(mean)
        except TypeError:
            pass
        else:
            mean = ConstantModel(constant)
        if not fit_mean:
            for name in mean.get_parameter_names():
                mean.freeze_parameter(name)

        super(GP, self).__init__([("kernel", kernel), ("mean", mean)])

    # -- members ----------------------------------------------------------------
    @property
    def solver(self):
        """One device-backed ``CholeskySolver`` per GP, created on first use."""
        if self._solver is None:
            self._solver = solver.CholeskySolver()
        return self._solver

    @property
    def mean(self):
        return self.models["mean"]

    @property
    def kernel(self):
        return self.models["kernel"]

    # -- life-cycle flags ----------------------------------------------------------
    @property
    def dirty(self):
        return ModelSet.dirty.fget(self) or not self._computed

    @dirty.setter
    def dirty(self, flag):
        self._computed = not flag
        ModelSet.dirty.fset(self, flag)

    @property
    def computed(self):
        return self._solver is not None and self.solver.computed() and not self.dirty

    # -- factorisation ---------------------------------------------------------------
    @staticmethod
    def _check_coordinates(t):
        t = np.atleast_1d(t)
        if np.any(np.diff(t) < 0.0):
            raise ValueError("the input coordinates must be sorted")
        if t.ndim > 1:
            raise ValueError("dimension mismatch")
        return t

    def compute(self, t, yerr=1.123e-12, check_sorted=True, A=None, U=None, V=None):
        """Factorise ``K = k(|t_i - t_j|) + diag(yerr^2 + jitter)`` (+ general terms).

        Raises ``ValueError`` for unsorted ``t`` and ``solver.LinAlgError`` for
        a matrix that is not positive definite.
        """
        t = self._check_coordinates(t) if check_sorted else np.atleast_1d(t)
        self._t = t
        # NB: like the reference (celerite.py:131) the buffer takes t's dtype.
        self._yerr = np.empty_like(t)
        self._yerr[:] = yerr
        a_real, c_real, a_comp, b_comp, c_comp, d_comp = self.kernel.coefficients
        dflt = _no_general()
        self._A = dflt[0] if A is None else A
        self._U = dflt[1] if U is None else U
        self._V = dflt[2] if V is None else V
        self.solver.compute(self.kernel.jitter,
                            a_real, c_real, a_comp, b_comp, c_comp, d_comp,
                            self._A, self._U, self._V, t, self._yerr ** 2)
        self.dirty = False

    def _recompute(self):
        if self.computed:
            return
        if self._t is None:
            raise RuntimeError("you must call 'compute' first")
        self.compute(self._t, self._yerr, check_sorted=False,
                     A=self._A, U=self._U, V=self._V)

    def _process_input(self, y):
        if self._t is None:
            raise RuntimeError("you must call 'compute' first")
        if len(self._t) != len(y):
            raise ValueError("dimension mismatch")
        return np.ascontiguousarray(y, dtype=float)

    # -- the likelihood ------------------------------------------------------------------
    def log_likelihood(self, y, _const=_LOG_2PI, quiet=False):
        """Marginal log-likelihood of ``y`` under the GP.

        ``quiet=True`` returns ``-inf`` instead of raising ``LinAlgError``.
        """
        y = self._process_input(y)
        resid = y - self.mean.get_value(self._t)
        if y.ndim == 1 and not self.computed:
            # the factorisation is about to run: let it fold resid^T K^-1 resid into the same pass over
            # the series (dot_solve below then returns it; any other vector takes the ordinary sweep)
            self.solver._hint_rhs(resid)
        try:
            self._recompute()
        except solver.LinAlgError:
            if quiet:
                return -np.inf
            raise
        except BaseException:
            # compute() consumes the hint; if the recompute failed before reaching it (a kernel or mean model
            # raising), withdraw it so that an unrelated later compute() does not fold a stale vector in
            self.solver._hint_rhs(np.empty(0))
            raise
        if y.ndim > 1:
            raise ValueError("dimension mismatch")
        logdet = self.solver.log_determinant()
        if not np.isfinite(logdet):
            return -np.inf
        value = -0.5 * (self.solver.dot_solve(resid) + logdet + len(y) * _const)
        return value if np.isfinite(value) else -np.inf

    def grad_log_likelihood(self, y, quiet=False):
        """Value and gradient w.r.t. :meth:`get_parameter_vector`.

        The forward-mode gradient of ``compute`` + ``dot_solve`` with respect to
        the coefficients runs on the device (``solver.has_autodiff()`` is True:
        csrc/grad_kernels.hip, the counterpart of solver.cpp:347-463, including
        its ``pi * log(N)`` constant).  The chain rule through the term
        parameters (``get_coeffs_jacobian`` / ``get_jitter_jacobian``, terms.py:197-215)
        needs no ``autograd`` for the built-in terms and their sums / products -- their
        formulas are evaluated on dual numbers (``terms._dual_coefficients``); a
        user-defined term needs it exactly as in the reference (``ImportError`` without).
        """
        if not solver.has_autodiff():
            raise RuntimeError("celerite must be compiled with autodiff "
                               "support to use the gradient methods")
        if not self.kernel.vector_size:
            return self.log_likelihood(y, quiet=quiet), np.empty(0)

        y = self._process_input(y)
        if y.ndim > 1:
            raise ValueError("dimension mismatch")
        resid = y - self.mean.get_value(self._t)
        a_real, c_real, a_comp, b_comp, c_comp, d_comp = self.kernel.coefficients
        try:
            value, grad = self.solver.grad_log_likelihood(
                self.kernel.jitter, a_real, c_real, a_comp, b_comp, c_comp, d_comp,
                self._A, self._U, self._V, self._t, resid, self._yerr ** 2)
        except solver.LinAlgError:
            if quiet:
                return -np.inf, np.zeros(self.vector_size)
            raise

        if self.kernel._has_coeffs:
            full = np.dot(self.kernel.get_coeffs_jacobian(), grad[1:])
        else:
            full = np.zeros(self.kernel.vector_size)
        if self.kernel._has_jitter:
            full += self.kernel.get_jitter_jacobian() * grad[0]
        if self.mean.vector_size:
            self._recompute()
            alpha = self.solver.solve(resid)
            full = np.append(full, np.dot(self.mean.get_gradient(self._t), alpha))
        return value, full

    # -- linear algebra with the factor ---------------------------------------------------
    def apply_inverse(self, y):
        """``K^-1 y`` (``K`` includes ``yerr^2`` and the jitter); shape (N, nrhs)."""
        self._recompute()
        return self.solver.solve(self._process_input(y))

    def dot(self, y, t=None, A=None, U=None, V=None, kernel=None, check_sorted=True):
        """``K y`` without the white-noise diagonal, in O(N)."""
        if kernel is None:
            kernel = self.kernel
        if t is not None:
            t = self._check_coordinates(t) if check_sorted else np.atleast_1d(t)
            dflt = _no_general()
            A = dflt[0] if A is None else A
            U = dflt[1] if U is None else U
            V = dflt[2] if V is None else V
        else:
            if not self.computed:
                raise RuntimeError("you must call 'compute' first")
            t, A, U, V = self._t, self._A, self._U, self._V
        a_real, c_real, a_comp, b_comp, c_comp, d_comp = kernel.coefficients
        return self.solver.dot(kernel.jitter, a_real, c_real, a_comp, b_comp, c_comp,
                               d_comp, A, U, V, t, np.ascontiguousarray(y, dtype=float))

    def predict(self, y, t=None, return_cov=True, return_var=False):
        """Conditional mean (and covariance / variance) at ``t`` given ``y``."""
        y = self._process_input(y)
        if y.ndim > 1:
            raise ValueError("dimension mismatch")
        if t is None:
            xs = self._t
        else:
            xs = np.ascontiguousarray(t, dtype=float)
            if xs.ndim > 1:
                raise ValueError("dimension mismatch")

        self._recompute()
        resid = y - self.mean.get_value(self._t)

        if t is None:
            alpha = self.solver.solve(resid).flatten()
            alpha = resid - (self._yerr ** 2 + self.kernel.jitter) * alpha
        elif not len(self._A):
            alpha = self.solver.predict(resid, xs)
        else:
            cross = self.get_matrix(xs, self._t)
            alpha = np.dot(cross, self.solver.solve(resid).flatten())

        mu = self.mean.get_value(xs) + alpha
        if not (return_var or return_cov):
            return mu

        cross = self.get_matrix(xs, self._t)
        crossT = np.ascontiguousarray(cross.T, dtype=np.float64)
        if return_var:
            var = -np.sum(crossT * self.apply_inverse(crossT), axis=0)
            var += self.kernel.get_value(0.0)
            return mu, var
        cov = self.kernel.get_value(xs[:, None] - xs[None, :])
        cov -= np.dot(cross, self.apply_inverse(crossT))
        return mu, cov

    def get_matrix(self, x1=None, x2=None, include_diagonal=None, include_general=None):
        """Dense covariance matrix (for checks and O(M^3) conditionals)."""
        if x1 is None and x2 is None:
            if self._t is None or not self.computed:
                raise RuntimeError("you must call 'compute' first")
            K = self.kernel.get_value(self._t[:, None] - self._t[None, :])
            if include_diagonal is None or include_diagonal:
                K[np.diag_indices_from(K)] += self._yerr ** 2 + self.kernel.jitter
            if (include_general is None or include_general) and len(self._A):
                K[np.diag_indices_from(K)] += self._A
                K += np.tril(np.dot(self._U.T, self._V), -1)
                K += np.triu(np.dot(self._V.T, self._U), 1)
            return K

        x1 = np.ascontiguousarray(x1, dtype=float)
        add_jitter = False
        if x2 is None:
            x2 = x1
            add_jitter = bool(include_diagonal)
        K = self.kernel.get_value(x1[:, None] - x2[None, :])
        if add_jitter:
            K[np.diag_indices_from(K)] += self.kernel.jitter
        return K

    # -- sampling -----------------------------------------------------------------------------
    def sample(self, size=None):
        """Draw from the prior: ``mean + L z`` with ``K = L L^T`` in O(N)."""
        self._recompute()
        shape = (len(self._t),) if size is None else (len(self._t), size)
        draws = self.solver.dot_L(np.random.randn(*shape))
        if size is None:
            return self.mean.get_value(self._t) + draws[:, 0]
        return self.mean.get_value(self._t)[None, :] + draws.T

    def sample_conditional(self, y, t=None, size=None, regularize=None):
        """Draw from the predictive distribution (O(M^3) in ``len(t)``)."""
        mu, cov = self.predict(y, t, return_cov=True)
        if regularize is not None:
            cov[np.diag_indices_from(cov)] += regularize
        return np.random.multivariate_normal(mu, cov, size=size)
