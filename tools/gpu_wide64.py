# -*- coding: utf-8 -*-
"""Widths 33..64 parallel in n (round 5): ms per evaluation of a wide plan by chunk count and first-chunk ratio, against the
sequential sweep (one chunk = one wave per problem), with the oracle on a few problems; and one long series through
CholeskySolver.  Log: profiles/r05d_wide64_chunks.txt"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_inputs
from celerite_amd import batch
import celerite_amd
from oracle import ref

N = 100000


def timed(plan, coeffs, reps=3):
    plan.set_coefficients(*coeffs); plan.log_likelihood()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        plan.set_coefficients(*coeffs); out = plan.log_likelihood()
        best = min(best, (time.perf_counter() - t0) * 1e3)
    return best, out


for JC, B in ((32, 256), (32, 64), (32, 512), (20, 256), (24, 128)):
    coeffs, t, diag, y = make_inputs(B, N, 0, JC, seed=B + JC, d_spread=True)
    plan = batch.BatchedGP(B, N, 0, JC)
    plan.set_series(t, diag, y)
    row = []
    base = None
    for nchunk, ratio in ((1, None), (0, None), (0, 1.5), (0, 2.3), (2, None), (4, None), (8, None), (16, None)):
        if B * max(nchunk, 1) > 4096:
            continue
        if ratio is not None:
            os.environ["CLR_WIDE_FIRST_RATIO64"] = str(ratio)
        else:
            os.environ.pop("CLR_WIDE_FIRST_RATIO64", None)
        plan.set_chunks(nchunk)
        ms, (ll, ld, q, st) = timed(plan, coeffs)
        k = {a: round(b, 2) for a, b in plan.run_timed(2)[1].items() if b > 0.02}
        if nchunk == 1:
            base = (ms, ld, q)
        lvl = np.bincount(plan.exact_levels(), minlength=3).tolist()
        row.append("chunks %s%s: %.2f ms (x%.2f) %s routes %s" % (plan.chunks, "" if ratio is None else " ratio %.1f" % ratio, ms, base[0] / ms,
                                                                 {a: round(b / 2, 2) for a, b in k.items()}, lvl))
        dev = max(np.max(np.abs(ld - base[1]) / np.abs(base[1])), np.max(np.abs(q - base[2]) / np.abs(base[2])))
        assert dev < 1e-10, dev
    os.environ.pop("CLR_WIDE_FIRST_RATIO64", None)
    l0, d0, q0, s0 = ref.batch_log_likelihood(0.0, *[c[:2] for c in coeffs], t[:2], diag[:2], y[:2])
    print("width %d B=%d: vs oracle %.1e / %.1e\n   %s" % (2 * JC, B, np.max(np.abs(base[1][:2] - d0) / np.abs(d0)),
                                                           np.max(np.abs(base[2][:2] - q0) / np.abs(q0)), "\n   ".join(row)), flush=True)
    plan.close()

# one long series through the object API
for JC in (20, 32):
    coeffs, t, diag, y = make_inputs(1, N, 0, JC, seed=5, d_spread=True)
    e, e2 = np.empty(0), np.empty((0, 0))
    args = (0.0,) + tuple(c[0] for c in coeffs) + (e, e2, e2, t[0], diag[0])
    r = ref.RefSolver(); t0 = time.perf_counter(); r.compute(*args); q0 = r.dot_solve(y[0]); cpu = (time.perf_counter() - t0) * 1e3
    out = []
    for env in (None, "1"):
        if env: os.environ["CLR_SOLVER_WIDE_CHUNKS"] = env
        s = celerite_amd.CholeskySolver()
        s.compute(*args); s.dot_solve(y[0])
        t0 = time.perf_counter()
        for _ in range(3):
            s.compute(*args); qq = s.dot_solve(y[0])
        out.append(((time.perf_counter() - t0) / 3 * 1e3, abs(s.log_determinant() - r.log_determinant()) / abs(r.log_determinant()), abs(qq - q0) / abs(q0)))
        os.environ.pop("CLR_SOLVER_WIDE_CHUNKS", None)
    print("object API width %d, N=%d: chunked %.2f ms (%.1e / %.1e), one chunk %.2f ms, CPU oracle %.1f ms" % (2 * JC, N, out[0][0], out[0][1], out[0][2], out[1][0], cpu), flush=True)
