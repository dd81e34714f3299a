# -*- coding: utf-8 -*-
"""Development probe for the GPU box: parity across widths / edge shapes, then a
chunk-count sweep with per-kernel HIP-event timings at the bench configuration."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import celerite_amd  # noqa: E402
from celerite_amd import batch  # noqa: E402
from oracle import ref  # noqa: E402

out = {}
print(batch.device_info(), flush=True)
rng = np.random.RandomState(7)


def mk(B, N, JR, JC, fam):
    if fam == "bench":
        t = np.sort(rng.rand(B, N), axis=1); sig = rng.uniform(0.1, 0.2, (B, N)); y = np.sin(t)
    else:
        t = np.sort(rng.uniform(0, 0.8 * N, (B, N)), axis=1); sig = rng.uniform(1.0, 1.5, (B, N)); y = rng.randn(B, N)
    ar = np.exp(1.0 + 0.1 * rng.randn(B, JR)); cr = np.exp(0.1 + 0.1 * rng.randn(B, JR))
    ac = np.exp(0.1 + 0.1 * rng.randn(B, JC)); bc = 0.3 * ac * rng.rand(B, JC)
    cc = np.exp(2.0 + 0.1 * rng.randn(B, JC)); dc = np.exp(1.6 + 0.1 * rng.randn(B, JC))
    return ar, cr, ac, bc, cc, dc, t, sig ** 2, y


worst = 0.0
for (JR, JC) in [(1, 0), (2, 0), (0, 1), (1, 1), (3, 0), (2, 1), (0, 2), (4, 0), (1, 2), (3, 1), (5, 0),
                 (2, 2), (0, 3), (6, 0), (4, 1), (1, 3), (3, 2), (5, 1), (7, 0), (2, 3), (0, 4), (4, 2), (6, 1), (8, 0)]:
    for fam in ("bench", "acc"):
        for N, nch in [(1, 0), (2, 0), (7, 3), (300, 0), (3000, 0), (3000, 64), (20000, 128)]:
            a = mk(5, N, JR, JC, fam)
            ll, ld, q, st = batch.batch_log_likelihood(*a, nchunk=nch)
            l0, d0, q0, s0 = ref.batch_log_likelihood(0.0, *a)
            e = max(np.max(np.abs(ld - d0) / np.abs(d0)), np.max(np.abs(q - q0) / np.abs(q0)))
            worst = max(worst, e)
            if e > 1e-11 or (st != s0).any():
                print("PARITY FAIL", JR, JC, fam, N, nch, e, st, s0, flush=True)
print("batch parity worst rel err: %.3e" % worst, flush=True)
out["batch_parity_worst"] = worst

# single-solver parity incl. general terms + wide kernels
np.random.seed(42)
t = np.sort(np.random.rand(500)); diag = np.random.uniform(0.1, 0.5, 500); b = np.random.randn(500)
U = np.vander(t - np.mean(t), 4).T; V = U * np.random.rand(4)[:, None]; A = np.sum(U * V, axis=0) + 1e-8
ar = np.array([1.5, 0.1, 0.6, 0.3, 0.8, 0.7]); cr = np.array([1.0, 0.3, 0.05, 0.01, 0.1, 0.2])
ac = np.array([1.0, 2.0]); bc = np.array([0.1, 0.5]); cc = np.array([1.0, 1.0]); dc = np.array([1.0, 1.0])
for gen in (False, True):
    g = (A, U, V) if gen else (np.empty(0), np.empty((0, 0)), np.empty((0, 0)))
    s = celerite_amd.CholeskySolver(); r = ref.RefSolver()
    s.compute(0.0, ar, cr, ac, bc, cc, dc, *g, t, diag); r.compute(0.0, ar, cr, ac, bc, cc, dc, *g, t, diag)
    B5 = np.random.randn(500, 5)
    print("solver general=%s: logdet %.2e dot_solve %.2e solve %.2e dot_L %.2e dot %.2e predict %s" % (
        gen, abs(s.log_determinant() - r.log_determinant()), abs(s.dot_solve(b) - r.dot_solve(b)),
        np.abs(s.solve(B5) - r.solve(B5)).max(), np.abs(s.dot_L(B5) - r.dot_L(B5)).max(),
        np.abs(s.dot(0.0, ar, cr, ac, bc, cc, dc, *g, t, B5) - r.dot(0.0, ar, cr, ac, bc, cc, dc, *g, t, B5)).max(),
        "n/a" if gen else "%.2e" % np.abs(s.predict(b, np.linspace(-0.1, 1.1, 77)) - r.predict(b, np.linspace(-0.1, 1.1, 77))).max()), flush=True)

# timing sweep at the bench configuration
B, N, JR, JC = 1024, 100000, 2, 3
t0 = time.time()
a = mk(B, N, JR, JC, "bench")
print("generated inputs in %.1fs" % (time.time() - t0), flush=True)
plan = batch.BatchedGP(B, N, JR, JC)
plan.set_series(a[6], a[7], a[8])
plan.set_coefficients(*a[:6])
res = {}
for layout in ("staged",):
    plan.set_layout(layout)
    for nch in (64, 128):
        plan.set_chunks(nch)
        plan.log_likelihood()  # warm
        tot, k = plan.run_timed(3, relayout_each_step=True)
        res["%s_%d" % (layout, nch)] = dict(ms=tot / 3, kernels={a: b / 3 for a, b in k.items()}, chunks=plan.chunks)
        print("layout=%s nchunk %4d L %5d: %.3f ms/step  (%s) -> %.0f loglik/s" % (
            layout, plan.chunks[0], plan.chunks[1], tot / 3,
            " ".join("%s %.3f" % (a, b / 3) for a, b in k.items()), B / (tot / 3) * 1e3), flush=True)
        ll, ld, q, st = plan.log_likelihood()
        l0, d0, q0, s0 = ref.batch_log_likelihood(0.0, *[x[:4] for x in a[:6]], a[6][:4], a[7][:4], a[8][:4])
        print("   parity (4 problems): logdet rel %.2e quad rel %.2e" % (
            np.max(np.abs(ld[:4] - d0) / np.abs(d0)), np.max(np.abs(q[:4] - q0) / np.abs(q0))), flush=True)
plan.set_layout("staged")
for nch in (64,):
    plan.set_chunks(nch)
    outs = {}
    for exact in (False, True):
        plan.set_exact(exact)
        plan.log_likelihood()
        tot, k = plan.run_timed(5)
        outs[exact] = plan.log_likelihood()
        print("exact=%d nchunk %3d: %.3f ms/step (%s) -> %.0f loglik/s ; problems replayed: %d" % (
            exact, plan.chunks[0], tot / 5, " ".join("%s %.3f" % (a, b / 5) for a, b in k.items()),
            B / (tot / 5) * 1e3, plan.exact_count()), flush=True)
        res["exact%d_%d" % (exact, nch)] = dict(ms=tot / 5, kernels={a: b / 5 for a, b in k.items()})
    print("   replay-free vs exact: logdet rel %.2e quad rel %.2e status equal %s" % (
        np.max(np.abs(outs[0][1] - outs[1][1]) / np.abs(outs[1][1])),
        np.max(np.abs(outs[0][2] - outs[1][2]) / np.abs(outs[1][2])), np.array_equal(outs[0][3], outs[1][3])), flush=True)
plan.set_exact(False)
plan.set_chunks(64)
tot, k = plan.run_timed(3, materialize=True)
fb = B * N * (3 * 8 + 1) * 8.0
print("materialize (staged, 64 chunks): %.3f ms/step kernels %s ; factor bytes %.2f GB -> replay write rate %.2f TB/s" % (
    tot / 3, {a: round(b / 3, 3) for a, b in k.items()}, fb / 1e9, fb / (k["replay"] / 3 * 1e-3) / 1e12), flush=True)
phi, u, W, D = plan.factor(3)
rs = ref.RefSolver()
rs.compute(0.0, *[x[3] for x in a[:6]], np.empty(0), np.empty((0, 0)), np.empty((0, 0)), a[6][3], a[7][3])
st = rs.state()
print("factor parity problem 3: phi %.2e u %.2e W %.2e D %.2e" % (
    np.abs(phi - st[4]).max(), np.abs(u - st[5]).max(), np.abs(W / st[6] - 1).max(), np.abs(D / st[7] - 1).max()), flush=True)
out["sweep"] = res
plan.set_chunks(64)
ll, ld, q, st = plan.log_likelihood()
l0, d0, q0, s0 = ref.batch_log_likelihood(0.0, *[x[:8] for x in a[:6]], a[6][:8], a[7][:8], a[8][:8])
print("N=1e5 parity (8 problems): logdet rel %.2e quad rel %.2e" % (
    np.max(np.abs(ld[:8] - d0) / np.abs(d0)), np.max(np.abs(q[:8] - q0) / np.abs(q0))), flush=True)
t0 = time.time(); ref.batch_log_likelihood(0.0, *[x[:4] for x in a[:6]], a[6][:4], a[7][:4], a[8][:4]); dt = time.time() - t0
print("CPU oracle: %.1f ms per log-likelihood (1 thread)" % (dt / 4 * 1e3), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "gpu_first.json"), "w"), indent=1, default=float)
