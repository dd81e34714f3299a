# -*- coding: utf-8 -*-
"""The scan algebra of celerite_amd/csrc/clr_core.h (summarize -> prefix ->
replay; both series layouts) instantiated on the HOST by tests/hostcheck and
compared with the oracle.  This checks the mathematics the GPU kernels execute
(same templates, same source) in the CPU-only run; the device build itself is
checked by the `-m gpu` tests.  Tolerance: 1e-11 relative (the bar is 1e-10)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle import ref
from _cases import synthetic, adversarial, coeffs_of

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hostcheck")
SO = os.path.join(HERE, "libhostcheck.so")
SHAPES = [(1, 0), (2, 0), (3, 0), (0, 1), (1, 1), (2, 1), (0, 2), (2, 2), (2, 3), (0, 4), (4, 2), (8, 0)]


@pytest.fixture(scope="module")
def lib():
    src = os.path.join(HERE, "hostcheck.cpp")
    core = os.path.join(os.path.dirname(HERE), "..", "celerite_amd", "csrc", "clr_core.h")
    if (not os.path.exists(SO) or os.path.getmtime(SO) < max(os.path.getmtime(src), os.path.getmtime(core))):
        subprocess.check_call(["g++", "-O2", "-std=c++14", "-shared", "-fPIC", "-ffp-contract=off",
                               "-mfma", "-o", SO, src])
    return C.CDLL(SO)


def run(lib, JR, JC, nchunk, case, interleaved, materialize=False, fast=True, exact=False):
    dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int)
    P = lambda a: np.ascontiguousarray(a, dtype=np.float64).ctypes.data_as(dp)
    B, N = case["t"].shape
    J = JR + 2 * JC
    ll, ld, q = np.empty(B), np.empty(B), np.empty(B)
    st = np.zeros(B, dtype=np.int32)
    used = np.zeros(B, dtype=np.int32)
    phi, u = np.zeros((B, max(N - 1, 1), J)), np.zeros((B, max(N - 1, 1), J))
    W, D = np.zeros((B, N, J)), np.zeros((B, N))
    jit = np.zeros(B)
    keep = [np.ascontiguousarray(case[k], dtype=np.float64) for k in
            ("a_real", "c_real", "a_comp", "b_comp", "c_comp", "d_comp", "t", "diag", "y")]
    rc = lib.hostcheck_batch(B, N, JR, JC, nchunk, P(jit), *[k.ctypes.data_as(dp) for k in keep[:6]],
                             keep[6].ctypes.data_as(dp), C.c_long(N), keep[7].ctypes.data_as(dp),
                             C.c_long(N), keep[8].ctypes.data_as(dp), C.c_long(N), int(materialize),
                             int(interleaved), int(fast), int(exact), P(ll), P(ld), P(q),
                             st.ctypes.data_as(ip), P(phi), P(u), P(W), P(D), used.ctypes.data_as(ip))
    assert rc == 0
    run.used_exact = used
    return ll, ld, q, st, (phi, u, W, D)


@pytest.mark.parametrize("JR,JC", SHAPES)
@pytest.mark.parametrize("family", ["bench", "accuracy"])
def test_scan_matches_oracle(lib, JR, JC, family):
    for N, nchunk in [(1, 1), (2, 1), (7, 3), (100, 1), (1000, 7), (1000, 64), (3000, 100)]:
        case = synthetic(3, N, JR, JC, family, seed=N + 10 * JR + JC)
        l0, d0, q0, s0 = ref.batch_log_likelihood(0.0, *coeffs_of(case), case["t"], case["diag"], case["y"])
        for inter, fast, exact in ((0, True, False), (1, True, False), (1, False, False), (1, True, True)):
            ll, ld, q, st, _ = run(lib, JR, JC, nchunk, case, inter, fast=fast, exact=exact)
            if not exact and N >= 1000 and nchunk <= N // 8:
                # positive-definite problems with chunks of >= 8 samples must be settled by
                # the replay-free path (chunk_update), not by the exact fallback
                assert not run.used_exact.any()
            assert np.array_equal(st, s0)
            assert np.max(np.abs(ld - d0) / np.abs(d0)) < 1e-11
            assert np.max(np.abs(q - q0) / np.abs(q0)) < 1e-11
            assert np.max(np.abs(ll - l0) / np.abs(l0)) < 1e-11


def test_materialised_factor_matches_oracle_state(lib):
    case = synthetic(2, 700, 2, 3, "bench", seed=9)
    _, _, _, _, (phi, u, W, D) = run(lib, 2, 3, 9, case, 1, materialize=True)
    for p in range(2):
        s = ref.RefSolver()
        s.compute(0.0, *coeffs_of(case, p), np.empty(0), np.empty((0, 0)), np.empty((0, 0)),
                  case["t"][p], case["diag"][p])
        _, N, J, _, rphi, ru, rW, rD = s.state()
        assert np.allclose(phi[p].T, rphi, rtol=1e-13, atol=0)
        assert np.allclose(u[p].T, ru, rtol=1e-12, atol=1e-15)
        assert np.allclose(W[p].T, rW, rtol=1e-9, atol=1e-12)
        assert np.allclose(D[p], rD, rtol=1e-11, atol=0)


def test_not_positive_definite_is_flagged_not_propagated(lib):
    case = synthetic(3, 600, 1, 0, "bench", seed=4)
    case["a_real"][1, 0] = -1.0  # tests/test_celerite.py:324-327
    case["diag"][1] = 0.0
    l0, d0, q0, s0 = ref.batch_log_likelihood(0.0, *coeffs_of(case), case["t"], case["diag"], case["y"])
    assert list(s0) == [0, 2, 0]
    for nchunk in (1, 5, 40):
        ll, ld, q, st, _ = run(lib, 1, 0, nchunk, case, 1)
        assert list(st) == [0, 2, 0]
        assert run.used_exact[1] == 1 or nchunk == 1  # flagged by the zero-start pivots -> exact replay
        assert ll[1] == -np.inf
        for p in (0, 2):
            assert abs(ld[p] - d0[p]) < 1e-11 * abs(d0[p]) and abs(q[p] - q0[p]) < 1e-11 * abs(q0[p])


def test_phase_sincos_and_log_product(lib):
    """The two device math helpers of clr_core.h, host-instantiated: the FMA
    Cody-Waite sincos (absolute error < 1 ulp(1) = 2.2e-16 for |x| < 1e9, the
    host-checked validity range) and the mantissa-product form of sum(log D)."""
    dp = C.POINTER(C.c_double)
    rng = np.random.RandomState(0)
    k = np.arange(-200000, 200000, 7, dtype=np.float64) * (np.pi / 2)
    x = np.concatenate([rng.uniform(-10, 10, 200000), rng.uniform(-1e3, 1e3, 200000),
                        rng.uniform(-1e6, 1e6, 200000), rng.uniform(-1e9, 1e9, 200000),
                        k, np.nextafter(k, 1e300), np.nextafter(k, -1e300),
                        [0.0, -0.0, 1e-300, 999999999.5]])
    s, c = np.empty_like(x), np.empty_like(x)
    lib.hostcheck_sincos(len(x), x.ctypes.data_as(dp), s.ctypes.data_as(dp), c.ctypes.data_as(dp))
    xl = x.astype(np.longdouble)
    err = max(np.max(np.abs(s - np.sin(xl))), np.max(np.abs(c - np.cos(xl))))
    assert float(err) < 2.3e-16

    lib.hostcheck_logprod.restype = C.c_double
    d = np.exp(rng.uniform(-5, 5, 100000))
    got = lib.hostcheck_logprod(len(d), d.ctypes.data_as(dp))
    want = float(np.sum(np.log(d.astype(np.longdouble))))
    assert abs(got - want) < 1e-15 * abs(want)
    for scale in (1e-300, 1e300):  # exponent range: no under/overflow of the running product
        d = np.full(5000, scale)
        assert abs(lib.hostcheck_logprod(5000, d.ctypes.data_as(dp)) - 5000 * np.log(scale)) < 1e-9
    z = np.array([2.0, 0.0, 3.0])
    assert lib.hostcheck_logprod(3, z.ctypes.data_as(dp)) == -np.inf  # D = 0: cholesky.h:208 gives -inf too


def test_indefinite_chunks_are_caught_by_the_inertia_check(lib):
    """A matrix that is NOT positive definite although every chunk's own block is:
    the failure only shows once a chunk is conditioned on the past.  Zero-start
    pivots stay positive, so only the certificate of chunk_update (or det(I + Jm P))
    can flag it; the problem must be routed to the exact replay and come out with
    the oracle's status."""
    rng = np.random.RandomState(12)
    N = 400
    t = np.sort(rng.uniform(0, 40, (1, N)), axis=1)
    # a_real = (+3, -2.9): k(0) > 0 and small blocks are PD, the long-range part is not
    case = dict(a_real=np.array([[3.0, -2.9]]), c_real=np.array([[0.05, 0.06]]),
                a_comp=np.empty((1, 0)), b_comp=np.empty((1, 0)), c_comp=np.empty((1, 0)),
                d_comp=np.empty((1, 0)), t=t, diag=np.full((1, N), 1e-3), y=np.sin(t))
    l0, d0, q0, s0 = ref.batch_log_likelihood(0.0, *coeffs_of(case), case["t"], case["diag"], case["y"])
    assert s0[0] == 2, "test kernel is expected to be indefinite"
    for nchunk in (4, 25, 50):
        ll, ld, q, st, _ = run(lib, 2, 0, nchunk, case, 1)
        assert st[0] == 2 and ll[0] == -np.inf
        assert run.used_exact[0] == 1


def test_adversarial_problems_keep_the_reference_status(lib):
    """Near-singular / indefinite problems (tests/_cases.adversarial): whatever the
    replay-free path decides, the status word (linalg_exception or not) must be the
    oracle's for every problem, chunking and width; values are compared only loosely
    because at these condition numbers the reference itself is cond * eps away from
    the exact answer."""
    n_bad = n_exact = n_total = 0
    for trial in range(60):
        JR, JC = SHAPES[trial % len(SHAPES)]
        N = (50, 200, 1000)[trial % 3]
        case = adversarial(4, N, JR, JC, seed=1000 + trial)
        l0, d0, q0, s0 = ref.batch_log_likelihood(0.0, *coeffs_of(case), case["t"], case["diag"], case["y"])
        n_bad += int((s0 != 0).sum())
        for nchunk in (max(2, N // 40), max(2, N // 8)):
            ll, ld, q, st, _ = run(lib, JR, JC, nchunk, case, 1)
            assert np.array_equal(st, s0), (trial, nchunk)
            n_total += 4
            n_exact += int(run.used_exact.sum())
            ok = (s0 == 0) & np.isfinite(d0) & np.isfinite(q0)
            if ok.any():
                assert np.max(np.abs(ld[ok] - d0[ok]) / (1 + np.abs(d0[ok]))) < 1e-5
                assert np.max(np.abs(q[ok] - q0[ok]) / (1 + np.abs(q0[ok]))) < 1e-4
    assert n_bad >= 10              # the family does contain indefinite problems
    assert 0 < n_exact < n_total    # and both routes are exercised


def test_scan_is_as_close_to_the_exact_answer_as_the_reference_recurrence(lib):
    """On near-singular problems the scan and the oracle differ by more than 1e-10 -- but
    so do the oracle (the reference's own recurrence) and the exact answer: both are
    cond * eps approximations.  Against a 60-digit dense factorisation (N = 50) the
    scan's error must stay within a small factor of the reference's."""
    from oracle import dense

    checked = 0
    for seed, (JR, JC) in ((1012, (4, 2)), (1031, (2, 1)), (1044, (2, 2)), (1003, (0, 2))):
        case = adversarial(4, 50, JR, JC, seed=seed)
        l0, d0, q0, s0 = ref.batch_log_likelihood(0.0, *coeffs_of(case), case["t"], case["diag"], case["y"])
        ll, ld, q, st, _ = run(lib, JR, JC, 6, case, 1)
        assert np.array_equal(st, s0)
        for b in range(4):
            if s0[b] != 0 or not (np.isfinite(d0[b]) and np.isfinite(q0[b])):
                continue
            co = [np.asarray(c[b]) for c in coeffs_of(case)]
            try:
                ld_x, q_x = dense.mp_exact_logdet_quad(*co, case["t"][b], case["diag"][b], case["y"][b])
            except Exception:   # not positive definite even in 60 digits
                continue
            e_ref = max(abs(d0[b] - ld_x) / abs(ld_x), abs(q0[b] - q_x) / abs(q_x))
            e_scan = max(abs(ld[b] - ld_x) / abs(ld_x), abs(q[b] - q_x) / abs(q_x))
            assert e_scan <= 20.0 * max(e_ref, 1e-13), (seed, b, e_scan, e_ref)
            checked += 1
    assert checked >= 6


@pytest.mark.parametrize("family", ["bench", "accuracy"])
def test_multilevel_prefix_matches_the_sequential_walk(lib, family):
    """Row h of the scope table: element o element composition (clr_core.h: compose_elements) and the
    multi-level prefix built from it -- compose groups bottom-up, walk the top level, fan the start
    states out -- must reproduce the plain walk over the chunks: same statuses, results within 1e-12 of
    it and within 1e-11 of the oracle, nothing pushed to the exact route."""
    try:
        for JR, JC in SHAPES:
            for N, nchunk in [(1000, 7), (1000, 64), (3000, 100)]:
                case = synthetic(3, N, JR, JC, family, seed=N + 10 * JR + JC)
                l0, d0, q0, s0 = ref.batch_log_likelihood(0.0, *coeffs_of(case), case["t"], case["diag"], case["y"])
                lib.hostcheck_set_prefix(0, 0)
                _, ld, q, st, _ = run(lib, JR, JC, nchunk, case, 1)
                for levels, g in [(1, 2), (1, 8), (2, 3), (3, 2)]:
                    lib.hostcheck_set_prefix(levels, g)
                    _, ld2, q2, st2, _ = run(lib, JR, JC, nchunk, case, 1)
                    assert np.array_equal(st2, s0) and not run.used_exact.any()
                    assert np.max(np.abs(ld2 - ld) / np.abs(ld)) < 1e-12
                    assert np.max(np.abs(q2 - q) / np.abs(q)) < 1e-12
                    assert np.max(np.abs(ld2 - d0) / np.abs(d0)) < 1e-11
                    assert np.max(np.abs(q2 - q0) / np.abs(q0)) < 1e-11
    finally:
        lib.hostcheck_set_prefix(0, 0)


def test_multilevel_prefix_keeps_the_reference_status_on_adversarial_problems(lib):
    try:
        for trial in range(36):
            JR, JC = SHAPES[trial % len(SHAPES)]
            N = (200, 1000)[trial % 2]
            case = adversarial(4, N, JR, JC, seed=1000 + trial)
            l0, d0, q0, s0 = ref.batch_log_likelihood(0.0, *coeffs_of(case), case["t"], case["diag"], case["y"])
            for levels, g in [(1, 4), (2, 3)]:
                lib.hostcheck_set_prefix(levels, g)
                ll, ld, q, st, _ = run(lib, JR, JC, max(2, N // 8), case, 1)
                assert np.array_equal(st, s0), (trial, levels, g)
    finally:
        lib.hostcheck_set_prefix(0, 0)


def test_prefix_planner(lib):
    """clr_core.h: plan_prefix -- the level structure of the multi-level prefix.  Invariants for any request, and
    the choices the measured time model makes (profiles/r03a_prefix_ab.txt, r03p_prefix_ab.txt): the headline shape
    (1024 problems x 64 chunks, width 8: throughput-bound, two waves per SIMD) takes ONE level of large groups,
    four times that batch keeps the plain walk, BASELINE config 1 (256 x 125..250 chunks, width 4) and a single long
    series go multi-level."""
    def plan(nchunk, levels=-1, g=0, B=1, J=8):
        out = np.zeros(8, dtype=np.int32)
        tm = C.c_double()
        lib.hostcheck_plan_prefix(nchunk, levels, g, B, J, out.ctypes.data_as(C.POINTER(C.c_int)), C.byref(tm))
        return int(out[0]), list(out[1:4]), list(out[4:8]), tm.value

    for nchunk in (1, 2, 3, 7, 13, 64, 125, 250, 1000, 4096):
        for levels, g in ((-1, 0), (0, 0), (1, 2), (1, 8), (2, 3), (3, 2), (3, 5)):
            lv, gs, ns, tm = plan(nchunk, levels, g, B=7, J=5)
            assert 0 <= lv <= 3 and ns[0] == nchunk and tm > 0
            for l in range(3):
                assert ns[l + 1] == (ns[l] + gs[l] - 1) // gs[l]
                assert gs[l] >= 2 if l < lv else gs[l] == 1
                if l < lv:
                    assert ns[l] >= 2 * gs[l]          # a level keeps at least two groups
    lv, gs, ns, tm = plan(64, B=1024, J=8)             # headline: one level (measured best: groups of 8)
    assert lv == 1 and 6 <= gs[0] <= 10, (lv, gs)
    assert plan(64, B=4096, J=8)[0] == 0               # throughput-bound: the walk
    assert plan(125, B=256, J=4)[0] >= 1 and plan(250, B=256, J=4)[0] >= 1
    lv, gs, ns, tm = plan(2048, B=1, J=8)
    assert lv >= 2 and ns[lv] <= 64 and tm < 0.2 * plan(2048, 0, 0, B=1, J=8)[3]


@pytest.mark.parametrize("JR,JC", [(2, 3), (0, 2), (1, 1), (4, 0)])
def test_warm_started_recurrence_forgets_or_is_caught(lib, JR, JC):
    """The claim behind warm_kernel / warm_check_kernel (clr_batch_kernels.h), on the host instantiation of
    replay_chunk: on the paper's accuracy family a chunk started from the ZERO state 64 samples early reaches the
    state the previous chunk ends in (mismatch <= 1e-11 at every boundary) and the sums equal the oracle's; with a
    warm-up that is too short, or on the bench family (which does not forget within a chunk), the boundary mismatch is
    orders of magnitude above the tolerance -- the check, not the warm-up heuristic, certifies the result."""
    dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int)

    def run_warm(case, nchunk, K):
        B, N = case["t"].shape
        keep = [np.ascontiguousarray(case[k], dtype=np.float64) for k in
                ("a_real", "c_real", "a_comp", "b_comp", "c_comp", "d_comp", "t", "diag", "y")]
        jit = np.zeros(B)
        ld, q, res = np.empty(B), np.empty(B), np.empty(B)
        fl = np.zeros(B, dtype=np.int32)
        rc = lib.hostcheck_warm(B, N, JR, JC, nchunk, K, jit.ctypes.data_as(dp), *[k.ctypes.data_as(dp) for k in keep],
                                ld.ctypes.data_as(dp), q.ctypes.data_as(dp), res.ctypes.data_as(dp), fl.ctypes.data_as(ip))
        assert rc == 0
        return ld, q, res, fl

    case = synthetic(3, 4000, JR, JC, "accuracy", seed=7 + JR)
    l0, d0, q0, s0 = ref.batch_log_likelihood(0.0, *coeffs_of(case), case["t"], case["diag"], case["y"])
    ld, q, res, fl = run_warm(case, 10, 64)
    assert (fl == 0).all() and np.max(res) <= 1e-11
    assert np.max(np.abs(ld - d0) / np.abs(d0)) < 1e-11 and np.max(np.abs(q - q0) / np.abs(q0)) < 1e-11
    ld, q, res, fl = run_warm(case, 10, 2)
    assert np.min(res) > 1e-6                       # two samples of warm-up: caught at the boundaries
    case = synthetic(3, 4000, JR, JC, "bench", seed=8 + JR)
    ld, q, res, fl = run_warm(case, 10, 128)
    assert np.min(res) > 1e-9                       # dense sampling: no forgetting within 128 samples


@pytest.fixture(scope="module")
def gradlib():
    src = os.path.join(HERE, "hostcheck_grad.cpp")
    so = os.path.join(HERE, "libhostcheck_grad.so")
    csrc = os.path.join(os.path.dirname(HERE), "..", "celerite_amd", "csrc")
    deps = [src, os.path.join(csrc, "clr_core.h"), os.path.join(csrc, "clr_grad_core.h")]
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(d) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++14", "-shared", "-fPIC", "-ffp-contract=off", "-mfma", "-o", so, src])
    return C.CDLL(so)


@pytest.mark.parametrize("JR,JC", [(1, 0), (2, 0), (0, 1), (1, 1), (2, 1), (0, 2), (2, 3), (3, 2)])
def test_chunk_parallel_gradient_against_the_dual_number_oracle(gradlib, JR, JC):
    """csrc/clr_grad_core.h on the host: tangents per (chunk, direction group) from a ZERO tangent state at the true
    start state of the chunk, the three riders of the base trajectory per chunk, the walk over the chunks -- equal to
    oracle/grad.py (the reference's loops on dual numbers, solver.cpp:347-463) for 1, 3 and 8 chunks: the split of a
    tangent into its zero-start part and the homogeneous propagation is exact, not an approximation."""
    from oracle import grad as ograd
    dp = C.POINTER(C.c_double)
    P = lambda a: np.ascontiguousarray(a, dtype=np.float64).ctypes.data_as(dp)
    rng = np.random.default_rng(100 + JR + 7 * JC)
    N = 240
    t = np.sort(rng.uniform(0, 0.5 * N, N))
    diag = rng.uniform(0.5, 1.0, N)
    y = rng.normal(size=N)
    ar, cr = rng.uniform(0.5, 1.5, JR), rng.uniform(0.05, 0.5, JR)
    ac, bc = rng.uniform(0.5, 1.5, JC), rng.uniform(-0.1, 0.1, JC)
    cc, dc = rng.uniform(0.05, 0.5, JC), rng.uniform(0.5, 3.0, JC)
    jitter = 0.05
    e, e2 = np.empty(0), np.empty((0, 0))
    v0, g0 = ograd.grad_log_likelihood(jitter, ar, cr, ac, bc, cc, dc, e, e2, e2, t, y, diag)
    gradlib.hostcheck_grad.argtypes = [C.c_int] * 4 + [C.c_double] + [dp] * 9 + [C.POINTER(C.c_double)] * 2 + [dp]
    for nchunk in (1, 3, 8):
        ld, q = C.c_double(), C.c_double()
        g = np.zeros(1 + 2 * JR + 4 * JC)
        rc = gradlib.hostcheck_grad(N, JR, JC, nchunk, jitter, P(ar), P(cr), P(ac), P(bc), P(cc), P(dc), P(t), P(diag),
                                    P(y), C.byref(ld), C.byref(q), P(g))
        assert rc == 0
        v = -0.5 * (q.value + ld.value + np.pi * np.log(N))     # the reference's constant, solver.cpp:415
        assert abs(v - v0) <= 1e-13 * abs(v0)
        assert np.max(np.abs(g - g0)) <= 1e-10 * np.max(np.abs(g0)), (nchunk, np.max(np.abs(g - g0)) / np.max(np.abs(g0)))


@pytest.mark.parametrize("JR,JC", [(1, 0), (0, 1), (1, 1), (2, 1), (2, 3), (3, 2)])
def test_reverse_mode_gradient_against_the_dual_number_oracle(gradlib, JR, JC):
    """csrc/clr_grad_core.h, reverse mode, on the host: the riders pass with its per-sample record, the adjoint walk
    backwards over the chunks, one reverse sweep per chunk for all partials.  Equal to oracle/grad.py; the adjoint a
    sweep arrives at for its chunk's first sample equals the one the walk predicted from the riders (1e-13: the two
    are independent computations of the same quantity); dense and sparse series; and the reason the states are stored
    every K steps: rebuilding them backwards over a long stretch amplifies rounding errors without bound."""
    from oracle import grad as ograd
    dp = C.POINTER(C.c_double)
    P = lambda a: np.ascontiguousarray(a, dtype=np.float64).ctypes.data_as(dp)
    rng = np.random.default_rng(300 + JR + 7 * JC)
    N = 400
    diag = rng.uniform(0.5, 1.0, N)
    y = rng.normal(size=N)
    ar, cr = rng.uniform(0.5, 1.5, JR), rng.uniform(0.05, 0.5, JR)
    ac, bc = rng.uniform(0.5, 1.5, JC), rng.uniform(-0.1, 0.1, JC)
    cc, dc = rng.uniform(0.05, 0.5, JC), rng.uniform(0.5, 3.0, JC)
    e, e2 = np.empty(0), np.empty((0, 0))
    gradlib.hostcheck_grad_reverse.argtypes = ([C.c_int] * 4 + [C.c_double] + [dp] * 9 + [C.POINTER(C.c_double)] * 2 +
                                               [dp, C.POINTER(C.c_double), C.c_int] + [C.POINTER(C.c_double)] * 2)

    def run(t, nchunk, K):   # K > 0: a state stored every K steps; 0: where the accumulated decay asks for one; < 0: none
        ld, q, mm, dr, sf = C.c_double(), C.c_double(), C.c_double(), C.c_double(), C.c_double()
        g = np.zeros(1 + 2 * JR + 4 * JC)
        rc = gradlib.hostcheck_grad_reverse(N, JR, JC, nchunk, 0.05, P(ar), P(cr), P(ac), P(bc), P(cc), P(dc), P(t),
                                            P(diag), P(y), C.byref(ld), C.byref(q), P(g), C.byref(mm), K, C.byref(dr),
                                            C.byref(sf))
        assert rc == 0
        return g, mm.value, dr.value, sf.value

    fractions = {}
    for span, nchunk, K in ((0.02, 3, -1), (0.02, 1, 50), (0.5, 5, 4), (0.5, 2, 1), (5.0, 4, 1),
                            (0.02, 2, 0), (0.5, 3, 0), (5.0, 4, 0), (50.0, 2, 0)):
        t = np.sort(rng.uniform(0, span * N, N))
        v0, g0 = ograd.grad_log_likelihood(0.05, ar, cr, ac, bc, cc, dc, e, e2, e2, t, y, diag)
        g, mismatch, drift, stored = run(t, nchunk, K)
        assert np.max(np.abs(g - g0)) <= 1e-10 * np.max(np.abs(g0)), (span, nchunk, K)
        assert mismatch <= 1e-12, (span, nchunk, K, mismatch)
        if K == 0:
            assert drift <= 1e-9, (span, drift)      # the adaptive rule keeps every rebuilt stretch inside the budget
            fractions[span] = stored
    # the adaptive rule stores a few states on a dense series and (almost) every one on a very sparse series
    assert fractions[0.02] <= fractions[0.5] < fractions[5.0] < fractions[50.0] <= 1.0, fractions
    assert fractions[0.02] < 0.02 and fractions[50.0] > 0.5, fractions
    # stored states at least `span` steps apart, the ones in between rebuilt FORWARDS from the stored state (GradStore::span):
    # same partials, same certificate, 1 / span of the states on a series that forgets between any two samples
    for span_steps in (2, 3, 4, 7):
        gradlib.hostcheck_grad_set_span(span_steps)
        try:
            for span, nchunk in ((0.02, 2), (0.5, 3), (5.0, 4), (50.0, 2), (50.0, 5)):
                t = np.sort(rng.uniform(0, span * N, N))
                v0, g0 = ograd.grad_log_likelihood(0.05, ar, cr, ac, bc, cc, dc, e, e2, e2, t, y, diag)
                g, mismatch, drift, stored = run(t, nchunk, 0)
                assert np.max(np.abs(g - g0)) <= 1e-10 * np.max(np.abs(g0)), (span_steps, span, nchunk)
                assert mismatch <= 1e-12 and drift <= 1e-9, (span_steps, span, nchunk, mismatch, drift)
                if span == 50.0:
                    assert 0.5 / span_steps <= stored <= 1.0 / span_steps + 0.02, (span_steps, stored)
        finally:
            gradlib.hostcheck_grad_set_span(1)
    # fewer slots than the rule asks for: the sweep reports it (infinite drift = a failed certificate, the problem is
    # then redone in forward mode by the library)
    gradlib.hostcheck_grad_set_slot_limit(3)
    try:
        g, mismatch, drift, stored = run(np.sort(rng.uniform(0, 5.0 * N, N)), 2, 0)
    finally:
        gradlib.hostcheck_grad_set_slot_limit(0)
    assert np.isinf(drift) and stored <= 2 * 3 / N + 1e-12
    # no stored states over 400 samples of a series that forgets: the reconstruction is lost, and the drift says so
    t = np.sort(rng.uniform(0, 2.0 * N, N))
    v0, g0 = ograd.grad_log_likelihood(0.05, ar, cr, ac, bc, cc, dc, e, e2, e2, t, y, diag)
    g, mismatch, drift, stored = run(t, 2, -1)
    assert not np.max(np.abs(g - g0)) <= 1e-6 * np.max(np.abs(g0))
    assert not drift <= 1e-6
