# -*- coding: utf-8 -*-
"""ctypes front-end of the CPU oracle (oracle/celerite_ref.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, ``__graft_entry__.smoke()`` and
the ``cpu_baseline`` leg of ``bench.py``; never by anything under
``celerite_amd/``.  ``RefSolver`` mirrors the method surface of the reference's
``celerite.solver.CholeskySolver`` (celerite/solver.cpp:241-663) so parity
tests can drive the oracle and the HIP product with the same calls.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libcelerite_ref.so")

REF_OK, REF_DIMENSION_MISMATCH, REF_LINALG, REF_NOT_COMPUTED = 0, 1, 2, 3


class RefLinAlgError(Exception):
    """celerite::linalg_exception (cpp/include/celerite/exceptions.h:32-36)."""


def build(force=False):
    """Compile the oracle in place with its Makefile (gcc only)."""
    if force or not os.path.exists(_LIB_PATH):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _LIB_PATH


def _load():
    build()
    try:
        lib = C.CDLL(_LIB_PATH)
        if hasattr(lib, "refq_factor_solve"):
            return lib
    except OSError:
        pass
    build(force=True)      # (a library from before celerite_ref_quad.c, or built for another machine)
    return C.CDLL(_LIB_PATH)


_lib = _load()
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


def _vec(a):
    a = np.ascontiguousarray(np.atleast_1d(a), dtype=np.float64)
    return a, a.ctypes.data_as(_dp), int(a.shape[0])


def _mat(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if a.ndim != 2:
        a = a.reshape((0, 0)) if a.size == 0 else np.atleast_2d(a)
    return a, a.ctypes.data_as(_dp), int(a.shape[0]), int(a.shape[1])


_lib.ref_create.restype = C.c_void_p
_lib.ref_destroy.argtypes = [C.c_void_p]
_lib.ref_computed.argtypes = [C.c_void_p]
_lib.ref_log_determinant.argtypes = [C.c_void_p, _dp]
_lib.ref_dot_solve.argtypes = [C.c_void_p, C.c_int, _dp, _dp]
_lib.ref_solve.argtypes = [C.c_void_p, C.c_int, C.c_int, _dp, _dp]
_lib.ref_dot_L.argtypes = [C.c_void_p, C.c_int, C.c_int, _dp, _dp]
_lib.ref_predict.argtypes = [C.c_void_p, C.c_int, _dp, C.c_int, _dp, _dp]
_COEFF_ARGS = [C.c_int, _dp] * 6 + [C.c_int, _dp] + [C.c_int, C.c_int, _dp] * 2
_lib.ref_compute.argtypes = [C.c_void_p, C.c_double] + _COEFF_ARGS + [C.c_int, _dp, C.c_int, _dp]
_lib.ref_dot.argtypes = [C.c_double] + _COEFF_ARGS + [C.c_int, _dp, C.c_int, C.c_int, _dp, _dp]
_lib.ref_batch_log_likelihood.argtypes = (
    [C.c_int] * 4 + [_dp] * 7 + [_dp, C.c_long] * 3 + [_dp, _dp, _dp, _ip, C.c_int]
)


def _raise(status):
    if status == REF_OK:
        return
    if status == REF_DIMENSION_MISMATCH:
        raise RuntimeError("dimension mismatch")
    if status == REF_LINALG:
        raise RefLinAlgError("failed to factorize or solve matrix")
    if status == REF_NOT_COMPUTED:
        raise RuntimeError("you must call 'compute' first")
    raise RuntimeError("oracle error %d" % status)


def _coeff_args(a_real, c_real, a_comp, b_comp, c_comp, d_comp, A, U, V):
    keep, args = [], []
    for v in (a_real, c_real, a_comp, b_comp, c_comp, d_comp, A):
        arr, ptr, n = _vec(v)
        keep.append(arr)
        args += [n, ptr]
    for m in (U, V):
        arr, ptr, r, c = _mat(m)
        keep.append(arr)
        args += [r, c, ptr]
    return keep, args


class RefSolver(object):
    """The oracle behind the reference's CholeskySolver method names."""

    def __init__(self):
        self._h = C.c_void_p(_lib.ref_create())

    def __del__(self):
        try:
            _lib.ref_destroy(self._h)
        except Exception:
            pass

    def compute(self, jitter, a_real, c_real, a_comp, b_comp, c_comp, d_comp,
                A, U, V, x, diag):
        keep, args = _coeff_args(a_real, c_real, a_comp, b_comp, c_comp, d_comp, A, U, V)
        xa, xp, nx = _vec(x)
        da, dp, nd = _vec(diag)
        _raise(_lib.ref_compute(self._h, float(jitter), *(args + [nx, xp, nd, dp])))

    def computed(self):
        return bool(_lib.ref_computed(self._h))

    def log_determinant(self):
        out = C.c_double()
        _raise(_lib.ref_log_determinant(self._h, C.byref(out)))
        return out.value

    def dot_solve(self, b):
        ba, bp, nb = _vec(np.asarray(b, dtype=float).reshape(-1))
        out = C.c_double()
        _raise(_lib.ref_dot_solve(self._h, nb, bp, C.byref(out)))
        return out.value

    def _colmajor(self, b):
        b = np.asarray(b, dtype=np.float64)
        if b.ndim == 1:
            b = b[:, None]
        return np.asfortranarray(b)

    def solve(self, b):
        bf = self._colmajor(b)
        x = np.empty_like(bf, order="F")
        _raise(_lib.ref_solve(self._h, bf.shape[0], bf.shape[1],
                              bf.ctypes.data_as(_dp), x.ctypes.data_as(_dp)))
        return x

    def dot_L(self, z):
        zf = self._colmajor(z)
        y = np.empty_like(zf, order="F")
        _raise(_lib.ref_dot_L(self._h, zf.shape[0], zf.shape[1],
                              zf.ctypes.data_as(_dp), y.ctypes.data_as(_dp)))
        return y

    def dot(self, jitter, a_real, c_real, a_comp, b_comp, c_comp, d_comp,
            A, U, V, x, b):
        keep, args = _coeff_args(a_real, c_real, a_comp, b_comp, c_comp, d_comp, A, U, V)
        xa, xp, nx = _vec(x)
        zf = self._colmajor(b)
        y = np.empty_like(zf, order="F")
        _raise(_lib.ref_dot(float(jitter), *(args + [nx, xp, zf.shape[0], zf.shape[1],
                                                     zf.ctypes.data_as(_dp),
                                                     y.ctypes.data_as(_dp)])))
        return y

    def predict(self, y, x):
        ya, yp, ny = _vec(y)
        xa, xp, nx = _vec(x)
        out = np.empty(nx)
        _raise(_lib.ref_predict(self._h, ny, yp, nx, xp, out.ctypes.data_as(_dp)))
        return out

    def state(self):
        """(computed, N, J, log_det, phi, u, W, D) as in solver.cpp:36-42."""

        class _S(C.Structure):
            _fields_ = [("computed", C.c_int), ("N", C.c_int), ("J", C.c_int),
                        ("log_det", C.c_double), ("phi", _dp), ("u", _dp),
                        ("W", _dp), ("D", _dp)]

        s = C.cast(self._h, C.POINTER(_S)).contents
        N, J = s.N, s.J
        if not s.computed:
            return (False, N, J, 0.0, None, None, None, None)

        def grab(p, rows, cols):
            n = rows * cols
            if n <= 0:
                return np.zeros((rows, max(cols, 0)))
            return np.ctypeslib.as_array(p, shape=(n,)).copy().reshape((cols, rows)).T

        return (True, N, J, s.log_det, grab(s.phi, J, N - 1), grab(s.u, J, N - 1),
                grab(s.W, J, N), np.ctypeslib.as_array(s.D, shape=(N,)).copy())


def load_native():
    """The same oracle source compiled ON THIS MACHINE with ``-O3 -march=native`` (oracle/Makefile, target ``native``;
    BASELINE.md section 3.1) -- for bench.py's cpu_baseline leg on the GPU box, whose host CPU is not the build
    container's.  Returns ``(library, flags)`` or ``(None, reason)`` when there is no compiler / the build fails."""
    path = os.path.join(_HERE, "libcelerite_ref_native.so")
    try:
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "native"], timeout=120,
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        lib = C.CDLL(path)
        lib.ref_batch_log_likelihood.argtypes = _lib.ref_batch_log_likelihood.argtypes
    except Exception as e:  # no gcc / make on the box, a build error, a timeout
        return None, repr(e)
    flags = "-O3 -march=native -fno-fast-math -ffp-contract=off"
    try:
        out = subprocess.check_output(["gcc", "-march=native", "-Q", "--help=target"], timeout=30, stderr=subprocess.DEVNULL).decode()
        arch = [ln.split()[-1] for ln in out.splitlines() if ln.strip().startswith("-march=")]
        if arch:
            flags += " (native = %s)" % arch[0]
    except Exception:
        pass
    return lib, flags


def batch_log_likelihood(jitter, a_real, c_real, a_comp, b_comp, c_comp, d_comp,
                         t, diag, y, nthreads=1, lib=None):
    """Oracle version of the batched entry point.

    Coefficients: (B, J_real) / (B, J_comp); ``t``, ``diag``, ``y``: (B, N) or
    (N,) for one series shared by every draw.  Returns
    ``(loglike, logdet, quad, status)``.
    """
    a_real = np.ascontiguousarray(np.atleast_2d(a_real), dtype=np.float64)
    B, J_real = a_real.shape
    c_real = np.ascontiguousarray(c_real, dtype=np.float64).reshape(B, J_real)
    a_comp = np.ascontiguousarray(a_comp, dtype=np.float64).reshape(B, -1)
    J_comp = a_comp.shape[1]
    b_comp = np.ascontiguousarray(b_comp, dtype=np.float64).reshape(B, J_comp)
    c_comp = np.ascontiguousarray(c_comp, dtype=np.float64).reshape(B, J_comp)
    d_comp = np.ascontiguousarray(d_comp, dtype=np.float64).reshape(B, J_comp)
    jitter = np.ascontiguousarray(np.broadcast_to(np.asarray(jitter, dtype=np.float64), (B,)))

    def series(a):
        a = np.ascontiguousarray(a, dtype=np.float64)
        if a.ndim == 1:
            return a, 0, a.shape[0]
        assert a.shape[0] == B
        return a, a.shape[1], a.shape[1]

    t, ts, N = series(t)
    diag, ds, _ = series(diag)
    y, ys, _ = series(y)
    ll = np.empty(B)
    ld = np.empty(B)
    q = np.empty(B)
    st = np.zeros(B, dtype=np.int32)
    p = lambda a: a.ctypes.data_as(_dp)
    (lib or _lib).ref_batch_log_likelihood(B, N, J_real, J_comp, p(jitter), p(a_real), p(c_real),
                                           p(a_comp), p(b_comp), p(c_comp), p(d_comp),
                                           p(t), ts, p(diag), ds, p(y), ys,
                                           p(ll), p(ld), p(q), st.ctypes.data_as(_ip), int(nthreads))
    return ll, ld, q, st


def quad_factor_solve(jitter, a_real, c_real, a_comp, b_comp, c_comp, d_comp, t, diag, y, want_factor=True):
    """ONE problem through the reference's recurrences carried in IEEE binary128 (oracle/celerite_ref_quad.c:
    cholesky.h:126-179, :236-260, :343-357 in the same operation order, __float128 from the double inputs).  Returns
    ``(W[J, N], D[N], x[N], logdet, quad)`` rounded to double -- "the truth" deviations are attributed with; ``W`` and
    ``D`` are ``None`` unless ``want_factor``.  Raises ``RefLinAlgError`` where the reference would (cholesky.h:176)."""
    ar, pa, JR = _vec(a_real)
    cr, pc, _ = _vec(c_real)
    ac, pac, JC = _vec(a_comp)
    bc, pbc, _ = _vec(b_comp)
    cc, pcc, _ = _vec(c_comp)
    dc, pdc, _ = _vec(d_comp)
    tt, pt, N = _vec(t)
    dd, pd, _ = _vec(diag)
    yy, py, _ = _vec(y)
    J = JR + 2 * JC
    W = np.empty(J * N) if want_factor else None
    D = np.empty(N) if want_factor else None
    x = np.empty(N)
    ld, q = C.c_double(), C.c_double()
    _lib.refq_factor_solve.restype = C.c_int
    _lib.refq_factor_solve.argtypes = [C.c_double, C.c_int, _dp, _dp, C.c_int, _dp, _dp, _dp, _dp, C.c_int, _dp, _dp, _dp,
                                       _dp, _dp, _dp, _dp, _dp]
    st = _lib.refq_factor_solve(float(jitter), JR, pa, pc, JC, pac, pbc, pcc, pdc, N, pt, pd, py,
                                W.ctypes.data_as(_dp) if want_factor else None, D.ctypes.data_as(_dp) if want_factor else None,
                                x.ctypes.data_as(_dp), C.byref(ld), C.byref(q))
    if st == REF_LINALG:
        raise RefLinAlgError("failed to factorize or solve matrix")
    if st != REF_OK:
        raise ValueError("dimension mismatch")
    return (W.reshape(N, J).T.copy() if want_factor else None), D, x, ld.value, q.value
