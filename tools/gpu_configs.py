# -*- coding: utf-8 -*-
"""All five BASELINE.json configurations on one MI355X next to the single-thread CPU oracle:
time per evaluation, log-likelihoods/s, parity on a sample."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from celerite_amd import GP, terms, batch  # noqa: E402
from oracle import ref  # noqa: E402


def mk(B, N, JR, JC, seed, dspread=False):
    rng = np.random.RandomState(seed)
    t = np.sort(rng.rand(B, N), axis=1); sig = rng.uniform(0.1, 0.2, (B, N)); y = np.sin(t)
    ar = np.exp(1.0 + 0.1 * rng.randn(B, JR)); cr = np.exp(0.1 + 0.1 * rng.randn(B, JR))
    ac = np.exp(0.1 + 0.1 * rng.randn(B, JC)); bc = np.zeros((B, JC))
    cc = np.exp(2.0 + 0.1 * rng.randn(B, JC))
    dc = np.exp(rng.uniform(0.0, 3.0, (B, JC))) if dspread else np.exp(1.6 + 0.1 * rng.randn(B, JC))
    return (ar, cr, ac, bc, cc, dc), t, sig ** 2, y


def best(fn, reps=5):
    fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return min(ts)


# config 1: single series N = 1000, 1 real + 1 SHO (width 3), the GP object API
rng = np.random.RandomState(0)
t = np.sort(rng.uniform(0, 50, 1000)); yerr = rng.uniform(0.1, 0.3, 1000); y = np.sin(t)
kernel = terms.RealTerm(0.1, 0.5) + terms.SHOTerm(log_S0=0.1, log_Q=1.0, log_omega0=0.5)
gp = GP(kernel)
tg = best(lambda: (gp.compute(t, yerr), gp.log_likelihood(y)))
co = kernel.coefficients
gen = (np.empty(0), np.empty((0, 0)), np.empty((0, 0)))
def cpu1():
    r = ref.RefSolver(); r.compute(0.0, *co, *gen, t, yerr ** 2); return r.dot_solve(y) + r.log_determinant()
tc = best(cpu1)
print("config 1  N=1000 width 3, GP.compute + log_likelihood (object API): GPU %.3f ms, CPU oracle %.3f ms" % (tg * 1e3, tc * 1e3), flush=True)

for name, B, N, JR, JC, dspread in [("config 2", 256, 10000, 2, 1, False), ("config 3", 1024, 100000, 2, 3, False),
                                    ("config 5", 256, 100000, 0, 16, True)]:
    co, t, d, y = mk(B, N, JR, JC, 3, dspread)
    plan = batch.BatchedGP(B, N, JR, JC)
    plan.set_series(t, d, y); plan.set_coefficients(*co)
    ll, ld, q, st = plan.log_likelihood()
    tot, k = plan.run_timed(5)
    S = 4
    t0 = time.time(); l0, d0, q0, s0 = ref.batch_log_likelihood(0.0, *[c[:S] for c in co], t[:S], d[:S], y[:S]); cpu = (time.time() - t0) / S
    print("%s  B=%d N=%d width %d: %.3f ms per batch (chunks %s) -> %.0f loglik/s ; CPU oracle %.2f ms each = %.1f/s per core -> %.0fx ; "
          "parity logdet %.1e quad %.1e" % (name, B, N, JR + 2 * JC, tot / 5, plan.chunks, B / (tot / 5) * 1e3, cpu * 1e3, 1 / cpu,
                                           B / (tot / 5) * 1e3 * cpu, np.max(np.abs(ld[:S] - d0) / np.abs(d0)), np.max(np.abs(q[:S] - q0) / np.abs(q0))), flush=True)
    plan.close()
print("config 4  = config 3 on 8 GPUs (batch sharded, no collective): run by the driver via bench.py --gpus 8", flush=True)
