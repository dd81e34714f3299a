# -*- coding: utf-8 -*-
"""Batched log-likelihoods: B independent (time series x hyper-parameter draw)
problems per call -- the data-parallel axis the reference does not have.

One evaluation of problem ``p`` is exactly ``GP.compute`` + ``GP.log_likelihood``
of the reference (celerite/celerite.py:103-219):
``-0.5 (r^T K_p^-1 r + log det K_p + N log 2 pi)``, with ``quiet=True``
semantics for matrices that are not positive definite (``status[p] == 2`` and
``-inf``).  The arithmetic runs as the chunked-scan HIP kernels of
``csrc/clr_core.h`` through ``clr_batch_*`` in ``include/celerite_hip.h``; this
file is plumbing (ctypes).  There is no CPU fallback: without an MI355X every
call raises ``RuntimeError``.
"""
import ctypes as C
import os

import numpy as np

__all__ = ["BatchedGP", "ShardedBatchedGP", "shard_bounds", "batch_log_likelihood", "batch_grad_log_likelihood",
           "kernel_coefficient_table", "kernel_coefficient_jacobian_table", "chain_gradient", "LIB_PATH"]

LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libcelerite_hip.so")

CLR_OK, CLR_NOT_POSITIVE_DEFINITE = 0, 2

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)
_lib = None


def _load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("libcelerite_hip.so is not built: run `make` at the repository "
                          "root; there is no pure-Python fallback")
    lib = C.CDLL(LIB_PATH)
    lib.clr_last_error.restype = C.c_char_p
    lib.clr_status_string.restype = C.c_char_p
    lib.clr_status_string.argtypes = [C.c_int]
    lib.clr_version.restype = C.c_char_p
    lib.clr_batch_create.restype = C.c_void_p
    lib.clr_batch_create.argtypes = [C.c_int] * 5
    lib.clr_batch_destroy.argtypes = [C.c_void_p]
    lib.clr_batch_set_series.argtypes = [C.c_void_p, _dp, C.c_long, _dp, C.c_long, _dp, C.c_long]
    lib.clr_batch_set_coefficients.argtypes = [C.c_void_p] + [_dp] * 7
    lib.clr_batch_set_chunks.argtypes = [C.c_void_p, C.c_int]
    lib.clr_batch_get_chunks.argtypes = [C.c_void_p, _ip, _ip]
    lib.clr_batch_enqueue.argtypes = [C.c_void_p, C.c_int]
    lib.clr_batch_synchronize.argtypes = [C.c_void_p]
    lib.clr_batch_get_results.argtypes = [C.c_void_p, _dp, _dp, _dp, _ip]
    lib.clr_batch_get_factor.argtypes = [C.c_void_p, C.c_int, _dp, _dp, _dp, _dp]
    lib.clr_batch_run_timed.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, _dp, _dp]
    lib.clr_batch_set_layout.argtypes = [C.c_void_p, C.c_int]
    lib.clr_batch_set_library_trig.argtypes = [C.c_void_p, C.c_int]
    lib.clr_batch_set_summarize_mode.argtypes = [C.c_void_p, C.c_int]
    lib.clr_batch_get_summarize_kernel.argtypes = [C.c_void_p, _ip]
    lib.clr_batch_fp32_probe.argtypes = [C.c_void_p, _dp, _dp, _dp]
    lib.clr_batch_set_profiling.argtypes = [C.c_void_p, C.c_int]
    lib.clr_batch_get_profile.argtypes = [C.c_void_p, _dp, _ip]
    lib.clr_batch_set_prefix_mode.argtypes = [C.c_void_p, C.c_int]
    lib.clr_batch_set_replay_source.argtypes = [C.c_void_p, C.c_int]
    lib.clr_batch_set_general.argtypes = [C.c_void_p, C.c_int, _dp, C.c_long, _dp, C.c_long, _dp, C.c_long]
    lib.clr_batch_set_warm_start.argtypes = [C.c_void_p, C.c_int, C.c_int]
    lib.clr_batch_get_warm_start.argtypes = [C.c_void_p] + [_ip] * 7
    lib.clr_batch_set_prefix_plan.argtypes = [C.c_void_p, C.c_int, C.c_int]
    lib.clr_batch_grad.argtypes = [C.c_void_p, _dp, _dp, _ip]
    lib.clr_batch_get_grad_fallbacks.argtypes = [C.c_void_p, _ip]
    lib.clr_batch_set_grad_mode.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double]
    lib.clr_batch_get_grad_info.argtypes = [C.c_void_p, _ip, _ip, _dp]
    lib.clr_batch_get_prefix_plan.argtypes = [C.c_void_p, _ip, _ip, _ip]
    lib.clr_batch_debug_get_starts.argtypes = [C.c_void_p, _dp]
    lib.clr_batch_debug_compose_check.argtypes = [C.c_void_p, C.c_int, _dp, _dp]
    lib.clr_batch_get_selection_bounds.argtypes = [C.c_void_p, _dp, _dp, _dp, _dp, _dp]
    lib.clr_batch_set_selection_bounds.argtypes = [C.c_void_p] + [C.c_double] * 4
    lib.clr_sharded_get_summarize_kernel.argtypes = [C.c_void_p, _ip]
    lib.clr_batch_set_exact.argtypes = [C.c_void_p, C.c_int]
    lib.clr_batch_get_exact_count.argtypes = [C.c_void_p, _ip]
    lib.clr_batch_get_exact_flags.argtypes = [C.c_void_p, _ip]
    lib.clr_batch_get_conditioning.argtypes = [C.c_void_p, _dp, _dp, _dp]
    lib.clr_batch_get_conditioning_chunkwise.argtypes = [C.c_void_p, _dp]
    lib.clr_batch_get_measured_error.argtypes = [C.c_void_p, _dp]
    lib.clr_batch_set_certificate.argtypes = [C.c_void_p, C.c_double, C.c_double]
    lib.clr_batch_set_certificate_gamma.argtypes = [C.c_void_p, C.c_double, C.c_double]
    lib.clr_shard_bounds.argtypes = [C.c_int, C.c_int, C.c_int, _ip, _ip]
    lib.clr_sharded_create.restype = C.c_void_p
    lib.clr_sharded_create.argtypes = [C.c_int] * 4 + [_ip, C.c_int]
    lib.clr_sharded_destroy.argtypes = [C.c_void_p]
    lib.clr_sharded_last_error.restype = C.c_char_p
    lib.clr_sharded_num_shards.argtypes = [C.c_void_p]
    lib.clr_sharded_get_shard.argtypes = [C.c_void_p, C.c_int, _ip, _ip, _ip]
    lib.clr_sharded_set_chunks.argtypes = [C.c_void_p, C.c_int]
    lib.clr_sharded_get_chunks.argtypes = [C.c_void_p, C.c_int, _ip, _ip]
    lib.clr_sharded_set_summarize_mode.argtypes = [C.c_void_p, C.c_int]
    lib.clr_sharded_set_series.argtypes = [C.c_void_p, _dp, C.c_long, _dp, C.c_long, _dp, C.c_long]
    lib.clr_sharded_set_coefficients.argtypes = [C.c_void_p] + [_dp] * 7
    lib.clr_sharded_enqueue.argtypes = [C.c_void_p]
    lib.clr_sharded_synchronize.argtypes = [C.c_void_p]
    lib.clr_sharded_get_results.argtypes = [C.c_void_p, _dp, _dp, _dp, _ip]
    lib.clr_sharded_evaluate.argtypes = [C.c_void_p] + [_dp] * 7 + [_dp, _dp, _dp, _ip]
    lib.clr_sharded_run_timed.argtypes = [C.c_void_p, C.c_int, _dp]
    lib.clr_device_info.argtypes = [C.c_char_p, C.c_size_t, _ip, C.POINTER(C.c_size_t)]
    lib.clr_set_device.argtypes = [C.c_int]
    _lib = lib
    return lib


def _check(status):
    if status == CLR_OK:
        return
    lib = _load()
    msg = lib.clr_status_string(status).decode()
    detail = lib.clr_last_error().decode()
    raise RuntimeError(msg + (": " + detail if detail else ""))


def set_option(key, value=None):
    """A tuning / cross-check switch of the library (``clr_set_option``, include/celerite_hip.h lists the keys);
    ``value=None`` removes it.  Environment variables of the same names count only under ``CLR_ALLOW_ENV=1``."""
    lib = _load()
    lib.clr_set_option.argtypes = [C.c_char_p, C.c_char_p]
    _check(lib.clr_set_option(key.encode(), None if value is None else str(value).encode()))


def get_option(key):
    lib = _load()
    lib.clr_get_option.argtypes = [C.c_char_p]
    lib.clr_get_option.restype = C.c_char_p
    v = lib.clr_get_option(key.encode())
    return None if v is None else v.decode()


class option(object):
    """``with batch.option("CLR_GRAD_SEQUENTIAL", 1): ...`` -- the switch is removed (or restored) on exit."""

    def __init__(self, key, value="1"):
        self.key, self.value = key, value

    def __enter__(self):
        self.old = get_option(self.key)
        set_option(self.key, self.value)
        return self

    def __exit__(self, *exc):
        set_option(self.key, self.old)
        return False


def device_count():
    return int(_load().clr_device_count())


def measure_fp64(waves_per_simd=2, iters=20000, device=None):
    """``(tflops, clock_mhz, cycles_per_fma)``: the fp64 FMA rate the device's vector ALUs sustain under full load, the
    shader clock during it and the SIMD cycles per issued FMA (``clr_device_measure_fp64``; a roofline measurement).
    ``device``: the calling thread's current device is switched to it first (default: left as it is)."""
    lib = _load()
    if device is not None:
        _check(lib.clr_set_device(int(device)))
    a, b, c = C.c_double(), C.c_double(), C.c_double()
    lib.clr_device_measure_fp64.argtypes = [C.c_int, C.c_int] + [C.POINTER(C.c_double)] * 3
    _check(lib.clr_device_measure_fp64(int(waves_per_simd), int(iters), C.byref(a), C.byref(b), C.byref(c)))
    return a.value, b.value, c.value


def device_synchronize():
    _check(_load().clr_device_synchronize())


def device_info():
    lib = _load()
    name = C.create_string_buffer(256)
    cus = C.c_int()
    mem = C.c_size_t()
    _check(lib.clr_device_info(name, 256, C.byref(cus), C.byref(mem)))
    return dict(name=name.value.decode(), compute_units=cus.value, hbm_bytes=mem.value)


def device_memory():
    """``(free_bytes, total_bytes)`` of the current device's HBM (``clr_device_memory``)."""
    f, t = C.c_size_t(), C.c_size_t()
    _check(_load().clr_device_memory(C.byref(f), C.byref(t)))
    return f.value, t.value


def _f64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None:
        a = a.reshape(shape)
    return a


def _ptr(a):
    return a.ctypes.data_as(_dp)


class BatchedGP(object):
    """Device-resident plan for B problems of N samples and a fixed kernel shape.

    Args:
        B, N: batch size and samples per series.
        J_real, J_comp: number of real / complex celerite terms
            (width ``J = J_real + 2 J_comp``: 1..8 run the chunked scan with one lane
            per (problem, chunk), 9..64 one wave per (problem, chunk)).
        device: GPU index (one process per GPU; shard the batch across ranks).
    """

    def __init__(self, B, N, J_real, J_comp, device=0):
        lib = _load()
        self.B, self.N, self.J_real, self.J_comp = int(B), int(N), int(J_real), int(J_comp)
        self.J = self.J_real + 2 * self.J_comp
        self._h = lib.clr_batch_create(self.B, self.N, self.J_real, self.J_comp, int(device))
        if not self._h:
            raise RuntimeError("clr_batch_create failed: " + lib.clr_last_error().decode())
        self._h = C.c_void_p(self._h)
        self._evaluate_fn = None

    def close(self):
        if getattr(self, "_h", None):
            _load().clr_batch_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- inputs -------------------------------------------------------------
    def set_series(self, t, diag, y):
        """``t``, ``diag`` (= yerr**2), ``y`` (mean already subtracted): each
        ``(B, N)`` or ``(N,)`` for one series shared by all problems.  ``t``
        must be sorted along the last axis (checked: GP.compute does the same,
        celerite.py:126-129)."""
        arrs, strides = [], []
        for a in (t, diag, y):
            a = _f64(a)
            if a.shape == (self.N,):
                strides.append(0)
            elif a.shape == (self.B, self.N):
                strides.append(self.N)
            else:
                raise ValueError("dimension mismatch")
            arrs.append(a)
        lib = _load()
        _check(lib.clr_batch_set_series(self._h, _ptr(arrs[0]), strides[0], _ptr(arrs[1]),
                                        strides[1], _ptr(arrs[2]), strides[2]))
        # sortedness from the device-side scan of the uploaded t (np.diff over 0.8 GB of times costs more than the
        # whole transfer): an unsorted batch is dropped again
        dtmin = C.c_double()
        lib.clr_batch_get_series_order.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
        _check(lib.clr_batch_get_series_order(self._h, C.byref(dtmin)))
        if dtmin.value < 0.0:
            lib.clr_batch_clear_series.argtypes = [C.c_void_p]
            _check(lib.clr_batch_clear_series(self._h))
            raise ValueError("the input coordinates must be sorted")

    def set_coefficients(self, a_real, c_real, a_comp, b_comp, c_comp, d_comp, jitter=0.0):
        """Coefficient tables ``(B, J_real)`` / ``(B, J_comp)``; ``jitter`` scalar or ``(B,)``."""
        try:
            blocks = [_f64(a_real, (self.B, self.J_real)), _f64(c_real, (self.B, self.J_real)),
                      _f64(a_comp, (self.B, self.J_comp)), _f64(b_comp, (self.B, self.J_comp)),
                      _f64(c_comp, (self.B, self.J_comp)), _f64(d_comp, (self.B, self.J_comp))]
        except ValueError:
            raise ValueError("dimension mismatch")
        jit = np.ascontiguousarray(np.broadcast_to(np.asarray(jitter, dtype=np.float64), (self.B,)))
        _check(_load().clr_batch_set_coefficients(self._h, _ptr(jit), *[_ptr(b) for b in blocks]))

    def set_general(self, A, U, V):
        """General semiseparable terms (``CholeskySolver.compute``'s ``A, U, V``; cholesky.h:65-72,148-152) for the
        batch: ``A`` ``(B, N)`` or ``(N,)``, ``U`` and ``V`` ``(B, J_general, N)`` or ``(J_general, N)`` (shared by
        all problems).  Empty ``U`` removes them.  Up to a total width of 64 the plan then runs the wave-per-(problem,
        chunk) kernels with the general rows as one more row class, above that the any-width sequential kernel
        (:meth:`set_general_route`)."""
        U, V, A = _f64(U), _f64(V), _f64(A)
        if U.size == 0:
            _check(_load().clr_batch_set_general(self._h, 0, None, 0, None, 0, None, 0))
            return
        if U.shape != V.shape or U.shape[-1] != self.N or U.ndim not in (2, 3):
            raise ValueError("dimension mismatch")
        JG = U.shape[-2]
        if U.ndim == 3 and U.shape[0] != self.B:
            raise ValueError("dimension mismatch")
        if A.shape not in ((self.N,), (self.B, self.N)):
            raise ValueError("dimension mismatch")
        _check(_load().clr_batch_set_general(self._h, JG, _ptr(A), self.N if A.ndim == 2 else 0,
                                             _ptr(U), JG * self.N if U.ndim == 3 else 0,
                                             _ptr(V), JG * self.N if V.ndim == 3 else 0))

    def set_general_route(self, route=-1):
        """Plans with general terms: -1 the wide kernels when the total width allows (default), 1 the any-width
        sequential kernel (one workgroup per problem; the cross-check)."""
        lib = _load()
        lib.clr_batch_set_general_route.argtypes = [C.c_void_p, C.c_int]
        _check(lib.clr_batch_set_general_route(self._h, int(route)))

    # -- tuning -------------------------------------------------------------
    def set_chunks(self, nchunk):
        _check(_load().clr_batch_set_chunks(self._h, int(nchunk)))

    @property
    def chunks(self):
        n, l = C.c_int(), C.c_int()
        _load().clr_batch_get_chunks(self._h, C.byref(n), C.byref(l))
        return n.value, l.value

    # -- evaluation -----------------------------------------------------------
    def enqueue(self, materialize=False):
        _check(_load().clr_batch_enqueue(self._h, int(bool(materialize))))

    def synchronize(self):
        _check(_load().clr_batch_synchronize(self._h))

    def results(self):
        ll = np.empty(self.B)
        ld = np.empty(self.B)
        q = np.empty(self.B)
        st = np.empty(self.B, dtype=np.int32)
        _check(_load().clr_batch_get_results(self._h, _ptr(ll), _ptr(ld), _ptr(q),
                                             st.ctypes.data_as(_ip)))
        return ll, ld, q, st

    def evaluate(self, a_real, c_real, a_comp, b_comp, c_comp, d_comp, jitter=0.0):
        """One optimiser / MCMC evaluation in ONE library call (``clr_batch_evaluate``): new coefficient tables in,
        ``(loglike, logdet, quad, status)`` of all B problems out -- ``set_coefficients`` + ``enqueue`` + ``results``
        without two of the three trips through ctypes.  Arrays that already are C-contiguous float64 of the right shape
        are passed as they are."""
        B, JR, JC = self.B, self.J_real, self.J_comp
        tabs = []
        for a, w in ((a_real, JR), (c_real, JR), (a_comp, JC), (b_comp, JC), (c_comp, JC), (d_comp, JC)):
            if not (isinstance(a, np.ndarray) and a.dtype == np.float64 and a.flags.c_contiguous and a.shape == (B, w)):
                try:
                    a = _f64(a, (B, w))
                except ValueError:
                    raise ValueError("dimension mismatch")
            tabs.append(a)
        jit = np.ascontiguousarray(np.broadcast_to(np.asarray(jitter, dtype=np.float64), (B,)))
        ll, ld, q, st = np.empty(B), np.empty(B), np.empty(B), np.empty(B, dtype=np.int32)
        fn = self._evaluate_fn
        if fn is None:
            fn = _load().clr_batch_evaluate
            fn.argtypes = [C.c_void_p] * 12      # (plain addresses: ndarray.ctypes.data is cheaper than data_as)
            self._evaluate_fn = fn
        _check(fn(self._h, jit.ctypes.data, *([a.ctypes.data for a in tabs] + [ll.ctypes.data, ld.ctypes.data, q.ctypes.data, st.ctypes.data])))
        return ll, ld, q, st

    def log_likelihood(self, materialize=False):
        """Evaluate all B problems; returns ``(loglike, logdet, quad, status)``."""
        self.enqueue(materialize)
        return self.results()

    def factor(self, p):
        """``(phi, u, W, D)`` of problem ``p`` after a materialising run, shaped
        like the reference's pickled state (solver.cpp:36-42)."""
        N, J = self.N, self.J
        phi = np.empty((N - 1, J))
        u = np.empty((N - 1, J))
        W = np.empty((N, J))
        D = np.empty(N)
        _check(_load().clr_batch_get_factor(self._h, int(p), _ptr(phi), _ptr(u), _ptr(W), _ptr(D)))
        return phi.T, u.T, W.T, D

    def solve(self, b=None):
        """``K_p^-1 b_p`` for every problem from the factor of the last materialising run (``clr_batch_solve``;
        ``CholeskySolver.solve``, cholesky.h:218-318, for B problems at once).  ``b``: ``(B, N)`` or ``(B, nrhs, N)``;
        ``None``: the plan's own ``y`` (no upload).  Returns an array of the same shape."""
        lib = _load()
        lib.clr_batch_solve.argtypes = [C.c_void_p, C.c_int, _dp, _dp]
        if b is None:
            x = np.empty((self.B, self.N))
            _check(lib.clr_batch_solve(self._h, 1, None, _ptr(x)))
            return x
        b = _f64(b)
        if b.ndim not in (2, 3) or b.shape[0] != self.B or b.shape[-1] != self.N:
            raise ValueError("dimension mismatch")
        nrhs = 1 if b.ndim == 2 else b.shape[1]
        x = np.empty(b.shape)
        _check(lib.clr_batch_solve(self._h, int(nrhs), _ptr(b), _ptr(x)))
        return x

    def predict(self, xs):
        """The conditional mean ``K_p(x*, t_p) K_p^-1 y_p`` of every problem at the prediction points ``xs`` --
        ``(M,)`` shared by all problems or ``(B, M)`` -- from the factor of the last materialising run
        (``clr_batch_predict``; ``CholeskySolver.predict``, cholesky.h:599-698, for B problems).  Returns ``(B, M)``."""
        lib = _load()
        lib.clr_batch_predict.argtypes = [C.c_void_p, C.c_int, _dp, C.c_long, _dp]
        xs = _f64(xs)
        if xs.ndim == 1:
            stride = 0
        elif xs.ndim == 2 and xs.shape[0] == self.B:
            stride = xs.shape[1]
        else:
            raise ValueError("dimension mismatch")
        M = xs.shape[-1]
        pred = np.empty((self.B, M))
        _check(lib.clr_batch_predict(self._h, int(M), _ptr(xs), stride, _ptr(pred)))
        return pred

    def dot_L(self, z):
        """``L_p z_p`` with ``K_p = L_p L_p^T`` for every problem from the factor of the last materialising run
        (``clr_batch_dot_L``; ``CholeskySolver.dot_L``, cholesky.h:409-431, for B problems at once).  ``z``: ``(B, N)``
        or ``(B, nrhs, N)``; returns an array of the same shape."""
        lib = _load()
        lib.clr_batch_dot_L.argtypes = [C.c_void_p, C.c_int, _dp, _dp]
        z = _f64(z)
        if z.ndim not in (2, 3) or z.shape[0] != self.B or z.shape[-1] != self.N:
            raise ValueError("dimension mismatch")
        nrhs = 1 if z.ndim == 2 else z.shape[1]
        y = np.empty(z.shape)
        _check(lib.clr_batch_dot_L(self._h, int(nrhs), _ptr(z), _ptr(y)))
        return y

    def dot(self, z):
        """``K_p z_p`` for every problem at the coefficients in force (``clr_batch_dot``; ``CholeskySolver.dot``,
        cholesky.h:441-596, for B problems at once -- the kernel matrix WITHOUT the observational variance, as
        ``GP.dot``).  ``z``: ``(B, N)`` or ``(B, nrhs, N)``; returns an array of the same shape."""
        lib = _load()
        lib.clr_batch_dot.argtypes = [C.c_void_p, C.c_int, _dp, _dp]
        z = _f64(z)
        if z.ndim not in (2, 3) or z.shape[0] != self.B or z.shape[-1] != self.N:
            raise ValueError("dimension mismatch")
        nrhs = 1 if z.ndim == 2 else z.shape[1]
        y = np.empty(z.shape)
        _check(lib.clr_batch_dot(self._h, int(nrhs), _ptr(z), _ptr(y)))
        return y

    def sample(self, size=None, mean=None, random=None):
        """Draws from every problem's prior ``N(mean_p, K_p)`` (``GP.sample``, celerite.py:422-451: ``mean + L n`` with
        standard normal ``n``) from the factor of the last materialising run.  ``size=None``: ``(B, N)``; else
        ``(B, size, N)``.  ``mean``: ``None``, a scalar, ``(N,)`` or ``(B, N)``."""
        rng = np.random if random is None else random
        shape = (self.B, self.N) if size is None else (self.B, int(size), self.N)
        y = self.dot_L(rng.standard_normal(shape))
        if mean is not None:
            m = np.asarray(mean, dtype=float)
            if m.ndim == 2 and size is not None:
                m = m[:, None, :]
            y += m
        return y

    def solve_device_ms(self):
        """Device time of the last :meth:`solve` (its kernels, without the host <-> HBM copies)."""
        lib = _load()
        ms = C.c_double()
        lib.clr_batch_get_solve_ms.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
        _check(lib.clr_batch_get_solve_ms(self._h, C.byref(ms)))
        return ms.value

    KERNEL_NAMES = ("relayout", "summarize", "prefix", "correct", "replay", "finalize")

    LAYOUTS = {"rowmajor": 0, "interleaved": 1, "staged": 2}

    def set_layout(self, layout="staged"):
        """How the kernels read the series: ``"staged"`` (default: coalesced tiles
        transposed through LDS, no extra pass), ``"interleaved"`` (a cached
        chunk-interleaved copy built by a transpose kernel when the series change)
        or ``"rowmajor"`` (direct, slow; for A/B measurements)."""
        _check(_load().clr_batch_set_layout(self._h, self.LAYOUTS.get(layout, layout)))

    PREFIX_MODES = {"single": 0, "walk": 1, "multilevel": 2}

    def set_prefix_mode(self, mode="multilevel", cooperative=None):
        """Prefix phase (chunk elements -> chunk start states): ``"multilevel"`` (default: groups of
        elements composed in parallel, the composed ones walked, start states fanned out;
        csrc/clr_prefix_kernels.h), ``"walk"`` (16 lanes per problem, chunk after chunk) or ``"single"``
        (one lane: the host-checked form, cross-check).  ``True`` / ``False`` mean walk / single
        (also accepted as the keyword ``cooperative`` of earlier releases)."""
        if cooperative is not None:
            mode = bool(cooperative)
        if isinstance(mode, bool):
            mode = 1 if mode else 0
        _check(_load().clr_batch_set_prefix_mode(self._h, int(self.PREFIX_MODES.get(mode, mode))))

    def set_prefix_plan(self, levels=-1, group=0):
        """Level structure of the multi-level prefix: ``levels`` levels of groups of ``group`` elements
        (``levels < 0``: chosen from the chunk count)."""
        _check(_load().clr_batch_set_prefix_plan(self._h, int(levels), int(group)))

    @property
    def prefix_plan(self):
        """``(levels, group sizes, element counts per level)`` of the prefix phase."""
        lv = C.c_int()
        g = (C.c_int * 3)()
        n = (C.c_int * 4)()
        _check(_load().clr_batch_get_prefix_plan(self._h, C.byref(lv), g, n))
        return lv.value, list(g)[:max(lv.value, 0)], list(n)[:lv.value + 1]

    def debug_starts(self):
        """Chunk start states of the last evaluation, ``(B, nchunk, J (J + 1) / 2 + J)`` (widths 1..8)."""
        out = np.empty((self.B, self.chunks[0], self.J * (self.J + 1) // 2 + self.J))
        _check(_load().clr_batch_debug_get_starts(self._h, _ptr(out)))
        return out

    def compose_check(self, group):
        """Cooperative composition kernel against the single-lane host-checked form on the last
        evaluation's chunk elements, in groups of ``group``: ``(largest relative difference, largest
        magnitude)``."""
        d, m = C.c_double(), C.c_double()
        _check(_load().clr_batch_debug_compose_check(self._h, int(group), C.byref(d), C.byref(m)))
        return d.value, m.value

    def selection_bounds(self):
        """``dict(tmax, dxmax, dmax, cmax, set_series_host_ms)``: what the kernel selection looks at."""
        v = [C.c_double() for _ in range(5)]
        _check(_load().clr_batch_get_selection_bounds(self._h, *[C.byref(x) for x in v]))
        return dict(zip(("tmax", "dxmax", "dmax", "cmax", "set_series_host_ms"), [x.value for x in v]))

    def set_warm_start(self, mode=-1, forced_warmup=0):
        """Warm-started plain recurrence for series that forget their past (``clr_batch_set_warm_start``):
        -1 auto, 0 off, 1 forced with ``forced_warmup`` steps.  Takes effect at the next
        :meth:`set_coefficients`."""
        _check(_load().clr_batch_set_warm_start(self._h, int(mode), int(forced_warmup)))

    def set_small_mode(self, mode=-1):
        """One-launch evaluation of short narrow problems (``clr_batch_set_small_mode``): -1 automatic, 0 off,
        1 whenever supported (widths 1..4, 512 <= N <= 32768)."""
        lib = _load()
        lib.clr_batch_set_small_mode.argtypes = [C.c_void_p, C.c_int]
        _check(lib.clr_batch_set_small_mode(self._h, int(mode)))

    def small_mode_active(self):
        """Whether the next evaluation runs as one launch (series and coefficients must be set)."""
        a = C.c_int()
        lib = _load()
        lib.clr_batch_get_small_mode.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        _check(lib.clr_batch_get_small_mode(self._h, C.byref(a)))
        return bool(a.value)

    def warm_start(self):
        """``dict(active, chunks, chunk_len, warmup_min, warmup_max, settled, fallbacks)``."""
        v = [C.c_int() for _ in range(7)]
        _check(_load().clr_batch_get_warm_start(self._h, *[C.byref(x) for x in v]))
        return dict(zip(("active", "chunks", "chunk_len", "warmup_min", "warmup_max", "settled", "fallbacks"),
                        [x.value for x in v]))

    def grad_log_likelihood(self):
        """``(value[B], grad[B, 1 + 2 J_real + 4 J_comp], status[B])`` at the coefficients in force, parallel in n
        (``clr_batch_grad``; the reference's conventions per problem, ``CholeskySolver.grad_log_likelihood``,
        solver.cpp:347-463)."""
        NG = 1 + 2 * self.J_real + 4 * self.J_comp
        value, grad, st = np.empty(self.B), np.empty((self.B, NG)), np.empty(self.B, dtype=np.int32)
        _check(_load().clr_batch_grad(self._h, _ptr(value), _ptr(grad), st.ctypes.data_as(_ip)))
        return value, grad, st

    def set_grad_mode(self, mode="reverse", stored_state_distance=0, drift_tolerance=0.0):
        """``"reverse"`` (default: one sweep for all partials), ``"forward"`` (one tangent per partial) or
        ``"reverse-direct-riders"`` (reverse mode with the riders accumulated along the trajectory instead of taken
        from the scan's elements; A/B runs); ``clr_batch_set_grad_mode``."""
        _check(_load().clr_batch_set_grad_mode(self._h, {"reverse": 0, "forward": 1, "reverse-direct-riders": 2}[mode], int(stored_state_distance),
                                               float(drift_tolerance)))

    def grad_info(self):
        """``dict(reverse, forward_reruns, drift_max)`` of the last :meth:`grad_log_likelihood`."""
        r, n, d = C.c_int(), C.c_int(), C.c_double()
        _check(_load().clr_batch_get_grad_info(self._h, C.byref(r), C.byref(n), C.byref(d)))
        return {"reverse": bool(r.value), "forward_reruns": n.value, "drift_max": d.value}

    def grad_fallbacks(self):
        """Problems of the last :meth:`grad_log_likelihood` that took the sequential gradient kernel."""
        n = C.c_int()
        _check(_load().clr_batch_get_grad_fallbacks(self._h, C.byref(n)))
        return n.value

    def set_replay_source(self, source=-1):
        """Series view of the replay pass behind the role-split summarize: 0 the chunk-interleaved copy,
        1 the row-major arrays staged through LDS, -1 auto (``clr_batch_set_replay_source``)."""
        _check(_load().clr_batch_set_replay_source(self._h, int(source)))

    def set_rescue(self, mode=-1):
        """Route-1 problems (ill-conditioned, checked chunked replay) re-planned as a small plan of their own with many
        short chunks (``clr_batch_set_rescue``): -1 automatic (chunks of >= 1024 samples), 0 the inline replay, 1 always."""
        lib = _load()
        lib.clr_batch_set_rescue.argtypes = [C.c_void_p, C.c_int]
        _check(lib.clr_batch_set_rescue(self._h, int(mode)))

    def rescue(self):
        """Of the last fetched evaluation: problems re-planned (negative: replayed inline), running total, the side
        plan's chunking."""
        lib = _load()
        last, total, nc, L = C.c_int(), C.c_long(), C.c_int(), C.c_int()
        lib.clr_batch_get_rescue.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_long), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        _check(lib.clr_batch_get_rescue(self._h, C.byref(last), C.byref(total), C.byref(nc), C.byref(L)))
        return {"last": last.value, "total": total.value, "chunks": (nc.value, L.value)}

    FACTOR_LAYOUTS = {"reference": 0, "lean": 1}

    def set_factor_layout(self, layout="reference"):
        """What a materialising run keeps in HBM (``clr_batch_set_factor_layout``): ``"reference"`` -- phi, u, W, D,
        ``8 N (3 J + 1)`` bytes per problem -- or ``"lean"`` -- W and D only, ``8 N (J + 1)`` bytes; phi and u are pure
        functions of the times and the coefficients (cholesky.h:127-147) and :meth:`factor` regenerates them."""
        lib = _load()
        lib.clr_batch_set_factor_layout.argtypes = [C.c_void_p, C.c_int]
        _check(lib.clr_batch_set_factor_layout(self._h, self.FACTOR_LAYOUTS.get(layout, layout)))

    def set_factor_refine(self, samples=64):
        """Samples at the head of every chunk a materialising run recomputes from the previous chunk's replayed end
        state (``clr_batch_set_factor_refine``; 0 switches the refinement off)."""
        lib = _load()
        lib.clr_batch_set_factor_refine.argtypes = [C.c_void_p, C.c_int]
        _check(lib.clr_batch_set_factor_refine(self._h, int(samples)))

    def factor_bytes(self):
        """Bytes of factor per problem in HBM under the layout and chunking in force."""
        lib = _load()
        n = C.c_size_t()
        lib.clr_batch_get_factor_bytes.argtypes = [C.c_void_p, C.POINTER(C.c_size_t)]
        _check(lib.clr_batch_get_factor_bytes(self._h, C.byref(n)))
        return int(n.value)

    def set_materialize_pipeline(self, groups=0, summarize_cus=0, summarize_streams=1):
        """Materialising runs as a pipeline over ``groups`` groups of problems (``clr_batch_set_materialize_pipeline``):
        the summarize of one group beside the replay of the previous one, on streams owning ``summarize_cus`` /
        the remaining compute units.  ``groups=0`` switches it off."""
        lib = _load()
        lib.clr_batch_set_materialize_pipeline.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        _check(lib.clr_batch_set_materialize_pipeline(self._h, int(groups), int(summarize_cus), int(summarize_streams)))

    def cu_census(self, which=0):
        """Distinct compute units per XCD a grid reaches on the plan's stream (0), the pipeline's summarize stream (1)
        or its replay stream (2) (``clr_batch_debug_cu_census``)."""
        lib = _load()
        out = (C.c_int * 8)()
        lib.clr_batch_debug_cu_census.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int)]
        _check(lib.clr_batch_debug_cu_census(self._h, int(which), out))
        return list(out)

    def set_exact(self, force=True):
        """Replay every problem step by step (the reference's recurrence) instead
        of settling it from the chunk summaries; for A/B runs and cross-checks."""
        _check(_load().clr_batch_set_exact(self._h, int(bool(force))))

    def exact_count(self):
        """Problems of the last (synchronised) run that needed the exact replay."""
        n = C.c_int()
        _check(_load().clr_batch_get_exact_count(self._h, C.byref(n)))
        return n.value

    def exact_flags(self):
        """Boolean mask: which problems of the last run went through the exact recurrence."""
        return self.exact_levels() != 0

    def exact_levels(self):
        """Per problem: 0 settled from the chunk summaries, 1 chunked replay (end states
        consistent with the scan), 2 truly sequential recurrence."""
        f = np.zeros(self.B, dtype=np.int32)
        _check(_load().clr_batch_get_exact_flags(self._h, f.ctypes.data_as(_ip)))
        return f

    def set_certificate(self, max_gamma_over_mu=1e7, max_residual=1e-11, max_gamma=None, max_gamma_error=None):
        """Routing of ill-conditioned problems (``clr_batch_set_certificate``,
        ``clr_batch_set_certificate_gamma``); a bound <= 0 switches that test off.  The two gamma bounds
        (defaults of a new plan: 1e4 and 3e-9) are only touched when one of them is passed."""
        _check(_load().clr_batch_set_certificate(self._h, float(max_gamma_over_mu), float(max_residual)))
        if max_gamma is not None or max_gamma_error is not None:
            _check(_load().clr_batch_set_certificate_gamma(self._h, float(1e4 if max_gamma is None else max_gamma),
                                                           float(3e-9 if max_gamma_error is None else max_gamma_error)))

    def conditioning(self):
        """``(gamma_max, mu_min)`` per problem of the last run: largest ``a_n / D_n`` over the
        zero-start pivots, smallest certificate pivot (see ``clr_batch_get_conditioning``)."""
        g, m, r = np.empty(self.B), np.empty(self.B), np.empty(self.B)
        _check(_load().clr_batch_get_conditioning(self._h, _ptr(g), _ptr(m), _ptr(r)))
        self.last_residual = r
        return g, m

    def measured_error(self):
        """Per problem: the largest measured relative error of the chunks' ``G = (I + P Jm)^-1 P``
        (``clr_batch_get_measured_error``; the routing tests ``gamma_max * eG_max``)."""
        e = np.empty(self.B)
        _check(_load().clr_batch_get_measured_error(self._h, _ptr(e)))
        return e

    def conditioning_chunkwise(self):
        r = np.empty(self.B)
        _check(_load().clr_batch_get_conditioning_chunkwise(self._h, _ptr(r)))
        return r

    def set_summarize_mode(self, mode=-1):
        """summarize kernel of widths 7, 8: 0 single wave, 1 two roles on two waves per
        SIMD, 2 the same with the decay factored out of the state on dense series, -1 auto
        (``clr_batch_set_summarize_mode``; csrc/clr_split_kernels.h)."""
        _check(_load().clr_batch_set_summarize_mode(self._h, int(mode)))

    def summarize_kernel(self):
        """Name of the summarize kernel the next evaluation runs."""
        k = C.c_int()
        _check(_load().clr_batch_get_summarize_kernel(self._h, C.byref(k)))
        return ("single wave", "role split", "role split, lazy decay")[k.value]

    def fp32_probe(self):
        """``(logdet, quad, ms)`` of the sequential sweep with a float state (widths 9..32;
        a measurement of the fp32 tolerance, ``clr_batch_fp32_probe``)."""
        ld, q = np.empty(self.B), np.empty(self.B)
        ms = C.c_double()
        _check(_load().clr_batch_fp32_probe(self._h, _ptr(ld), _ptr(q), C.byref(ms)))
        return ld, q, ms.value

    def set_library_trig(self, force=True):
        """Use the library (ocml) sincos instead of the FMA Cody-Waite routine
        (which is picked automatically when max|d| * max|t| < 1e9)."""
        _check(_load().clr_batch_set_library_trig(self._h, int(bool(force))))

    def set_profiling(self, on=True):
        """Bracket the kernels of every following :meth:`enqueue` with HIP events; ``on=2`` brackets the
        summarize (dominant) kernel only: two event records per evaluation instead of seven."""
        _check(_load().clr_batch_set_profiling(self._h, 2 if on == 2 else int(bool(on))))

    def profile(self):
        """``({kernel name: summed ms}, evaluations recorded)`` since :meth:`set_profiling`."""
        k = (C.c_double * 6)()
        n = C.c_int()
        _check(_load().clr_batch_get_profile(self._h, k, C.byref(n)))
        return dict(zip(self.KERNEL_NAMES, [k[i] for i in range(6)])), n.value

    def run_timed(self, steps, materialize=False, relayout_each_step=True):
        """``steps`` back-to-back evaluations bracketed by HIP events on the
        plan's stream.  Returns ``(total_ms, {kernel name: summed ms})``."""
        tot = C.c_double()
        k = (C.c_double * 6)()
        _check(_load().clr_batch_run_timed(self._h, int(bool(materialize)), int(steps),
                                           int(bool(relayout_each_step)), C.byref(tot), k))
        return tot.value, dict(zip(self.KERNEL_NAMES, [k[i] for i in range(6)]))


def shard_bounds(total, nshards, shard):
    """``[lo, hi)`` of ``shard`` when ``total`` problems are cut into ``nshards``
    contiguous slices (``clr_shard_bounds``; pure host arithmetic, needs no GPU)."""
    lo, hi = C.c_int(), C.c_int()
    if _load().clr_shard_bounds(int(total), int(nshards), int(shard), C.byref(lo), C.byref(hi)) != CLR_OK:
        raise ValueError("bad shard arguments")
    return lo.value, hi.value


class ShardedBatchedGP(object):
    """The batch axis over several GPUs: ``len(devices)`` contiguous shards, one
    :class:`BatchedGP`-like plan and one host thread per shard, no collective
    (problems are independent: cholesky.h:703-706).  ``devices`` defaults to every
    visible GPU; a device may be listed more than once (shards sharing a GPU), which
    is how the sharding is tested on one GPU.  The kernels are selected once for the whole
    batch (batch-wide maxima of the series and coefficients), so results do not depend on the
    sharding bit for bit when the chunk count is the same (:meth:`set_chunks`; the automatic
    choice looks at the shard size) and the warm-started recurrence is not in play (it adapts per
    plan: see :meth:`set_warm_start`)."""

    def __init__(self, B, N, J_real, J_comp, devices=None):
        lib = _load()
        if devices is None:
            devices = list(range(device_count()))
        devices = [int(d) for d in devices]
        if not devices:
            raise RuntimeError("no gfx950 (MI355X) device is visible; libcelerite_hip has no CPU path")
        self.B, self.N, self.J_real, self.J_comp = int(B), int(N), int(J_real), int(J_comp)
        arr = (C.c_int * len(devices))(*devices)
        h = lib.clr_sharded_create(self.B, self.N, self.J_real, self.J_comp, arr, len(devices))
        if not h:
            raise RuntimeError("clr_sharded_create failed: " + lib.clr_sharded_last_error().decode())
        self._h = C.c_void_p(h)

    def _ok(self, status):
        if status != CLR_OK:
            lib = _load()
            raise RuntimeError(lib.clr_status_string(status).decode() + ": " +
                               lib.clr_sharded_last_error().decode())

    def close(self):
        if getattr(self, "_h", None):
            _load().clr_sharded_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def shards(self):
        """``[(device, lo, hi), ...]``"""
        lib = _load()
        out = []
        for s in range(lib.clr_sharded_num_shards(self._h)):
            d, lo, hi = C.c_int(), C.c_int(), C.c_int()
            lib.clr_sharded_get_shard(self._h, s, C.byref(d), C.byref(lo), C.byref(hi))
            out.append((d.value, lo.value, hi.value))
        return out

    def set_chunks(self, nchunk):
        self._ok(_load().clr_sharded_set_chunks(self._h, int(nchunk)))

    def set_summarize_mode(self, mode=-1):
        self._ok(_load().clr_sharded_set_summarize_mode(self._h, int(mode)))

    def set_rescue(self, mode=-1):
        """``clr_batch_set_rescue`` on every shard.  Side plan or inline replay of route-1 problems, and the side plan's
        chunk count, follow their number in the WHOLE batch: bit-identical under any sharding."""
        lib = _load()
        lib.clr_sharded_set_rescue.argtypes = [C.c_void_p, C.c_int]
        self._ok(lib.clr_sharded_set_rescue(self._h, int(mode)))

    def rescued(self):
        """Problems of the last fetched evaluation that took the checked route outside the main pass, over all shards."""
        lib = _load()
        n = C.c_int()
        lib.clr_sharded_get_rescue.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        self._ok(lib.clr_sharded_get_rescue(self._h, C.byref(n)))
        return n.value

    def set_certificate(self, max_gamma_over_mu=1e7, max_residual=1e-11, max_gamma=None, max_gamma_error=None):
        """``BatchedGP.set_certificate`` on every shard (the routing bounds of ill-conditioned problems)."""
        lib = _load()
        lib.clr_sharded_set_certificate.argtypes = [C.c_void_p] + [C.c_double] * 4
        touch = max_gamma is not None or max_gamma_error is not None
        self._ok(lib.clr_sharded_set_certificate(self._h, float(max_gamma_over_mu), float(max_residual),
                                                 float(1e4 if max_gamma is None else max_gamma) if touch else -1.0,
                                                 float(3e-9 if max_gamma_error is None else max_gamma_error) if touch else -1.0))

    def set_warm_start(self, mode=-1, forced_warmup=0):
        """``clr_batch_set_warm_start`` on every shard.  Activation (half of the problems of the WHOLE batch eligible) and
        the adaptation of the warm-up lengths are decided once for the batch: the same route -- the same bits -- under
        any sharding; ``mode=0`` switches the warm start off."""
        lib = _load()
        lib.clr_sharded_set_warm_start.argtypes = [C.c_void_p, C.c_int, C.c_int]
        self._ok(lib.clr_sharded_set_warm_start(self._h, int(mode), int(forced_warmup)))

    def summarize_kernel(self):
        """The summarize kernel ALL shards run (resolved once for the whole batch)."""
        k = C.c_int()
        self._ok(_load().clr_sharded_get_summarize_kernel(self._h, C.byref(k)))
        return ("single wave", "role split", "role split, lazy decay")[k.value] if k.value >= 0 else "shards disagree"

    def set_series(self, t, diag, y):
        arrs, strides = [], []
        for a in (t, diag, y):
            a = _f64(a)
            if a.shape == (self.N,):
                strides.append(0)
            elif a.shape == (self.B, self.N):
                strides.append(self.N)
            else:
                raise ValueError("dimension mismatch")
            arrs.append(a)
        lib = _load()
        self._ok(lib.clr_sharded_set_series(self._h, _ptr(arrs[0]), strides[0], _ptr(arrs[1]),
                                            strides[1], _ptr(arrs[2]), strides[2]))
        dtmin = C.c_double()
        lib.clr_sharded_get_series_order.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
        self._ok(lib.clr_sharded_get_series_order(self._h, C.byref(dtmin)))
        if dtmin.value < 0.0:       # (the device-side scans of the shards: see BatchedGP.set_series)
            lib.clr_sharded_clear_series.argtypes = [C.c_void_p]
            self._ok(lib.clr_sharded_clear_series(self._h))
            raise ValueError("the input coordinates must be sorted")

    def _coeff_blocks(self, a_real, c_real, a_comp, b_comp, c_comp, d_comp, jitter):
        try:
            blocks = [_f64(a_real, (self.B, self.J_real)), _f64(c_real, (self.B, self.J_real)),
                      _f64(a_comp, (self.B, self.J_comp)), _f64(b_comp, (self.B, self.J_comp)),
                      _f64(c_comp, (self.B, self.J_comp)), _f64(d_comp, (self.B, self.J_comp))]
        except ValueError:
            raise ValueError("dimension mismatch")
        jit = np.ascontiguousarray(np.broadcast_to(np.asarray(jitter, dtype=np.float64), (self.B,)))
        return jit, blocks

    def set_coefficients(self, a_real, c_real, a_comp, b_comp, c_comp, d_comp, jitter=0.0):
        jit, blocks = self._coeff_blocks(a_real, c_real, a_comp, b_comp, c_comp, d_comp, jitter)
        self._ok(_load().clr_sharded_set_coefficients(self._h, _ptr(jit), *[_ptr(b) for b in blocks]))

    def grad_log_likelihood(self):
        """``(value[B], grad[B, 1 + 2 J_real + 4 J_comp], status[B])`` at the coefficients in force: every shard's
        plan gradient concurrently (``clr_sharded_grad``)."""
        NG = 1 + 2 * self.J_real + 4 * self.J_comp
        value, grad, st = np.empty(self.B), np.empty((self.B, NG)), np.empty(self.B, dtype=np.int32)
        lib = _load()
        lib.clr_sharded_grad.argtypes = [C.c_void_p, _dp, _dp, _ip]
        self._ok(lib.clr_sharded_grad(self._h, _ptr(value), _ptr(grad), st.ctypes.data_as(_ip)))
        return value, grad, st

    def enqueue(self):
        self._ok(_load().clr_sharded_enqueue(self._h))

    def synchronize(self):
        self._ok(_load().clr_sharded_synchronize(self._h))

    def _out(self):
        return np.empty(self.B), np.empty(self.B), np.empty(self.B), np.empty(self.B, dtype=np.int32)

    def results(self):
        ll, ld, q, st = self._out()
        self._ok(_load().clr_sharded_get_results(self._h, _ptr(ll), _ptr(ld), _ptr(q), st.ctypes.data_as(_ip)))
        return ll, ld, q, st

    def log_likelihood(self):
        """Evaluate all B problems with the coefficients set last; returns ``(loglike, logdet, quad,
        status)`` (as :meth:`BatchedGP.log_likelihood`)."""
        self.enqueue()
        return self.results()

    def evaluate(self, a_real, c_real, a_comp, b_comp, c_comp, d_comp, jitter=0.0):
        """One optimiser / MCMC evaluation: new coefficients in, ``(loglike, logdet,
        quad, status)`` of all B problems out."""
        jit, blocks = self._coeff_blocks(a_real, c_real, a_comp, b_comp, c_comp, d_comp, jitter)
        ll, ld, q, st = self._out()
        self._ok(_load().clr_sharded_evaluate(self._h, _ptr(jit), *([_ptr(b) for b in blocks] +
                                              [_ptr(ll), _ptr(ld), _ptr(q), st.ctypes.data_as(_ip)])))
        return ll, ld, q, st

    def materialize(self):
        """A materialising evaluation on every shard (``clr_sharded_materialize``): ``(loglike, logdet, quad, status)``
        as :meth:`log_likelihood`, and every shard's factor left in its HBM for :meth:`solve`, :meth:`dot_L`,
        :meth:`sample` and :meth:`predict`."""
        ll, ld, q, st = self._out()
        lib = _load()
        lib.clr_sharded_materialize.argtypes = [C.c_void_p, _dp, _dp, _dp, _ip]
        self._ok(lib.clr_sharded_materialize(self._h, _ptr(ll), _ptr(ld), _ptr(q), st.ctypes.data_as(_ip)))
        return ll, ld, q, st

    def _rhs(self, b):
        b = _f64(b)
        if b.ndim not in (2, 3) or b.shape[0] != self.B or b.shape[-1] != self.N:
            raise ValueError("dimension mismatch")
        return b, (1 if b.ndim == 2 else b.shape[1])

    def solve(self, b=None):
        """``K_p^-1 b_p`` for every problem (as :meth:`BatchedGP.solve`), every shard on its slice."""
        lib = _load()
        lib.clr_sharded_solve.argtypes = [C.c_void_p, C.c_int, _dp, _dp]
        if b is None:
            x = np.empty((self.B, self.N))
            self._ok(lib.clr_sharded_solve(self._h, 1, None, _ptr(x)))
            return x
        b, nrhs = self._rhs(b)
        x = np.empty(b.shape)
        self._ok(lib.clr_sharded_solve(self._h, int(nrhs), _ptr(b), _ptr(x)))
        return x

    def dot_L(self, z):
        """``L_p z_p`` for every problem (as :meth:`BatchedGP.dot_L`), every shard on its slice."""
        lib = _load()
        lib.clr_sharded_dot_L.argtypes = [C.c_void_p, C.c_int, _dp, _dp]
        z, nrhs = self._rhs(z)
        y = np.empty(z.shape)
        self._ok(lib.clr_sharded_dot_L(self._h, int(nrhs), _ptr(z), _ptr(y)))
        return y

    sample = BatchedGP.sample

    def dot(self, z):
        """``K_p z_p`` for every problem (as :meth:`BatchedGP.dot`), every shard on its slice."""
        lib = _load()
        lib.clr_sharded_dot.argtypes = [C.c_void_p, C.c_int, _dp, _dp]
        z, nrhs = self._rhs(z)
        y = np.empty(z.shape)
        self._ok(lib.clr_sharded_dot(self._h, int(nrhs), _ptr(z), _ptr(y)))
        return y

    def predict(self, xs):
        """The conditional mean of every problem at ``xs`` (``(M,)`` shared or ``(B, M)``), as :meth:`BatchedGP.predict`."""
        lib = _load()
        lib.clr_sharded_predict.argtypes = [C.c_void_p, C.c_int, _dp, C.c_long, _dp]
        xs = _f64(xs)
        if xs.ndim == 1:
            stride = 0
        elif xs.ndim == 2 and xs.shape[0] == self.B:
            stride = xs.shape[1]
        else:
            raise ValueError("dimension mismatch")
        M = xs.shape[-1]
        pred = np.empty((self.B, M))
        self._ok(lib.clr_sharded_predict(self._h, int(M), _ptr(xs), stride, _ptr(pred)))
        return pred

    def run_timed(self, steps):
        """``steps`` evaluations on every shard concurrently; per-shard HIP-event ms."""
        n = _load().clr_sharded_num_shards(self._h)
        ms = np.zeros(n)
        self._ok(_load().clr_sharded_run_timed(self._h, int(steps), _ptr(ms)))
        return ms


def batch_log_likelihood(a_real, c_real, a_comp, b_comp, c_comp, d_comp, t, diag, y,
                         jitter=0.0, device=0, nchunk=0):
    """One-shot batched evaluation; see :class:`BatchedGP`."""
    a_real = np.atleast_2d(_f64(a_real))
    B, J_real = a_real.shape
    a_comp = _f64(a_comp).reshape(B, -1)
    N = np.asarray(t).shape[-1]
    plan = BatchedGP(B, N, J_real, a_comp.shape[1], device=device)
    try:
        if nchunk:
            plan.set_chunks(nchunk)
        plan.set_series(t, diag, y)
        plan.set_coefficients(a_real, c_real, a_comp, b_comp, c_comp, d_comp, jitter)
        return plan.log_likelihood()
    finally:
        plan.close()


def batch_grad_log_likelihood(a_real, c_real, a_comp, b_comp, c_comp, d_comp, t, diag, y, jitter=0.0, device=0):
    """Value and coefficient gradient of B problems at once (``clr_batch_grad_log_likelihood``): returns
    ``(value[B], grad[B, 1 + 2 J_real + 4 J_comp], status[B])`` with the reference's conventions per
    problem (``CholeskySolver.grad_log_likelihood``, solver.cpp:347-463; its ``pi log N`` constant)."""
    lib = _load()
    a_real = np.atleast_2d(_f64(a_real))
    B, JR = a_real.shape
    a_comp = _f64(a_comp).reshape(B, -1)
    JC = a_comp.shape[1]
    blocks = [a_real, _f64(c_real, (B, JR)), a_comp, _f64(b_comp, (B, JC)), _f64(c_comp, (B, JC)), _f64(d_comp, (B, JC))]
    N = np.asarray(t).shape[-1]
    arrs, strides = [], []
    for a in (t, diag, y):
        a = _f64(a)
        strides.append(0 if a.shape == (N,) else N)
        if a.shape not in ((N,), (B, N)):
            raise ValueError("dimension mismatch")
        arrs.append(a)
    jit = np.ascontiguousarray(np.broadcast_to(np.asarray(jitter, dtype=np.float64), (B,)))
    NG = 1 + 2 * JR + 4 * JC
    value, grad, st = np.empty(B), np.empty((B, NG)), np.empty(B, dtype=np.int32)
    lib.clr_batch_grad_log_likelihood.argtypes = ([C.c_int] * 4 + [_dp] * 7 + [_dp, C.c_long] * 3 +
                                                  [_dp, _dp, _ip, C.c_int])
    _check(lib.clr_batch_grad_log_likelihood(B, N, JR, JC, _ptr(jit), *[_ptr(b) for b in blocks],
                                             _ptr(arrs[0]), strides[0], _ptr(arrs[1]), strides[1],
                                             _ptr(arrs[2]), strides[2], _ptr(value), _ptr(grad),
                                             st.ctypes.data_as(_ip), int(device)))
    return value, grad, st


def kernel_coefficient_table(kernel, parameter_vectors):
    """Coefficient tables for many hyper-parameter draws of one ``terms.Term``.

    ``parameter_vectors``: ``(B, kernel.vector_size)``.  Returns
    ``(a_real, c_real, a_comp, b_comp, c_comp, d_comp, jitter)`` ready for
    :meth:`BatchedGP.set_coefficients`.  The kernel's parameters are restored.
    Every draw must give the same number of real / complex terms.
    """
    saved = kernel.get_parameter_vector()
    rows, jit = [], []
    try:
        for p in np.atleast_2d(parameter_vectors):
            kernel.set_parameter_vector(p)
            rows.append([np.array(b, dtype=np.float64) for b in kernel.coefficients])
            jit.append(float(kernel.jitter))
    finally:
        kernel.set_parameter_vector(saved)
    shapes = set(tuple(len(b) for b in r) for r in rows)
    if len(shapes) != 1:
        raise ValueError("the draws do not share one (J_real, J_comp) shape")
    blocks = [np.array([r[i] for r in rows]).reshape(len(rows), -1) for i in range(6)]
    return tuple(blocks) + (np.array(jit),)


def kernel_coefficient_jacobian_table(kernel, parameter_vectors):
    """The chain rule's other half for :func:`kernel_coefficient_table`: for every draw the Jacobian of the
    coefficients with respect to the kernel's (unfrozen) parameters.

    Returns ``(jac, jitter_jac)``: ``jac[b, p, c]`` = d coefficient ``c`` / d parameter ``p`` with the coefficients
    in the order of the batched gradient's columns 1.. (``a_real, c_real, a_comp, b_comp, c_comp, d_comp``, each
    block contiguous: ``Term.get_coeffs_jacobian``, terms.py:206-215) and ``jitter_jac[b, p]`` = d jitter / d
    parameter (``get_jitter_jacobian``, :197-204).  Built-in terms need no autograd (their formulas are evaluated
    on dual numbers, ``terms._dual_coefficients``)."""
    saved = kernel.get_parameter_vector()
    jac, jit = [], []
    try:
        for p in np.atleast_2d(parameter_vectors):
            kernel.set_parameter_vector(p)
            if kernel._has_coeffs:
                jac.append(np.asarray(kernel.get_coeffs_jacobian(), dtype=np.float64))
            else:
                jac.append(np.zeros((len(p), 0)))
            jit.append(np.asarray(kernel.get_jitter_jacobian(), dtype=np.float64) if kernel._has_jitter
                       else np.zeros(len(p)))
    finally:
        kernel.set_parameter_vector(saved)
    return np.array(jac), np.array(jit)


def chain_gradient(grad, jac, jitter_jac):
    """``d loglike / d parameters`` of every draw, ``[B, P]``, from the batched coefficient gradient ``grad``
    (``[B, 1 + 2 J_real + 4 J_comp]``: column 0 the jitter partial, as ``BatchedGP.grad_log_likelihood`` returns
    it) and the tables of :func:`kernel_coefficient_jacobian_table` (what ``GP.grad_log_likelihood`` does for one
    problem, celerite.py:286-305)."""
    grad = np.asarray(grad, dtype=np.float64)
    return np.einsum("bpc,bc->bp", jac, grad[:, 1:]) + jitter_jac * grad[:, :1]
