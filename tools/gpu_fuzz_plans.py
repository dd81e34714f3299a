# -*- coding: utf-8 -*-
"""Randomised cross-check of the batched plans against the CPU oracle: random (B, N, J_real, J_comp) over widths 1 .. 128,
both synthetic families, shared or per-problem series, a few indefinite problems mixed in; log-likelihood / log det /
quadratic form / status of every problem, on every third case a materialising run with the batched solve and the batched
dot_L (either factor layout on narrow plans), and on every case of width <= 64 the batched dot (before OR after the
evaluation: it must not depend on one having run) -- the last two also through a sharded plan on every sixth case.
Usage: gpu_fuzz_plans.py [cases] [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from celerite_amd import batch
from oracle import ref
from _cases import synthetic, coeffs_of

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
worst = {}
bad = 0
t0 = time.time()
E, E2 = np.empty(0), np.empty((0, 0))
for k in range(cases):
    width = int(rng.choice([1, 2, 3, 4, 5, 6, 7, 8, 9, 12, 16, 17, 24, 32, 33, 40, 48, 64, 65, 100, 128]))
    JC = int(rng.randint(0, width // 2 + 1))
    JR = width - 2 * JC
    N = int(rng.choice([1, 2, 7, 64, 255, 256, 257, 1000, 1023, 1024, 1025, 2048, 4000, 10000, 30000]))
    B = int(rng.choice([1, 2, 3, 7, 33, 64, 65]))
    if width > 32 and N * B > 400000:
        N = 4000
    if N * B * max(width, 8) ** 2 > 3e9:
        B = max(1, int(3e9 / (N * max(width, 8) ** 2)))
    family = "bench" if rng.randint(2) else "accuracy"
    case = synthetic(B, max(N, 2), JR, JC, family, seed=5000 + k)
    for key in ("t", "diag", "y"):
        case[key] = case[key][:, :N].copy()
    case["diag"] = case["diag"] + 0.02
    nbad = 0
    if JR and B >= 3 and rng.randint(3) == 0:
        case["a_real"] = np.array(case["a_real"], copy=True)
        case["a_real"][B // 2, :] = -7.0
        case["diag"][B // 2] = 0.0
        nbad = 1
    l0, d0, q0, s0 = ref.batch_log_likelihood(0.0, *coeffs_of(case), case["t"], case["diag"], case["y"])
    plan = batch.BatchedGP(B, N, JR, JC)
    try:
        lean = width <= 8 and rng.randint(2) == 1
        if lean:
            plan.set_factor_layout("lean")
        plan.set_series(case["t"], case["diag"], case["y"])
        plan.set_coefficients(*coeffs_of(case))
        zz = rng.randn(B, N)
        pd = int(rng.randint(B))
        dot_first = width <= 64 and rng.randint(2) == 0
        if dot_first:
            kz = plan.dot(zz)
        ll, ld, q, st = plan.log_likelihood()
        tag = (B, N, JR, JC, family, plan.chunks)
        if width <= 64:
            if not dot_first:
                kz = plan.dot(zz)
            want = ref.RefSolver().dot(0.0, *coeffs_of(case, pd), E, E2, E2, case["t"][pd], zz[pd])[:, 0]
            dv = np.max(np.abs(kz[pd] - want)) / max(np.max(np.abs(want)), 1e-300)
            if not (dv <= worst.get("batched dot", (0.0,))[0]):
                worst["batched dot"] = (float(dv), tag)
            if not (dv <= 1e-9):
                bad += 1
                print("ABOVE 1e-9: batched dot %.2e %s first=%s" % (dv, tag, dot_first), flush=True)
        if not np.array_equal(st != 0, s0 != 0):
            bad += 1
            print("STATUS differs:", tag, st, s0, flush=True)
        ok = (s0 == 0) & (st == 0)
        # (an indefinite FIRST pivot is never tested -- cholesky.h:176 starts at n = 1 -- so a series of one sample with a
        #  negative diagonal has status 0 and a NaN log det on both sides)
        if not np.array_equal(np.isnan(ld[ok]), np.isnan(d0[ok])):
            bad += 1
            print("NaN pattern differs:", tag, ld, d0, flush=True)
        ok &= np.isfinite(d0) & np.isfinite(q0)
        devs = {}
        if ok.any():
            devs["logdet"] = np.max(np.abs(ld[ok] - d0[ok]) / np.maximum(np.abs(d0[ok]), 1e-300))
            devs["quad"] = np.max(np.abs(q[ok] - q0[ok]) / np.abs(q0[ok]))
            if width <= 64 and max(devs["logdet"], devs["quad"]) > 1e-11:
                # attribution: the same recurrence carried in binary128 (oracle/celerite_ref_quad.c) -- device vs truth
                # beside sequential double oracle vs truth, on the problem that deviates most
                rel = np.where(ok, np.maximum(np.abs(ld - d0) / np.maximum(np.abs(d0), 1e-300), np.abs(q - q0) / np.maximum(np.abs(q0), 1e-300)), 0.0)
                pw = int(np.argmax(rel))
                try:
                    _, _, _, ldq, qq = ref.quad_factor_solve(0.0, *coeffs_of(case, pw), case["t"][pw], case["diag"][pw], case["y"][pw], want_factor=False)
                    print("ATTRIBUTION %s problem %d: log det device-truth %.1e, oracle-truth %.1e; quad device-truth %.1e, oracle-truth %.1e%s" % (
                        tag, pw, abs(ld[pw] - ldq) / abs(ldq), abs(d0[pw] - ldq) / abs(ldq), abs(q[pw] - qq) / abs(qq), abs(q0[pw] - qq) / abs(qq),
                        "  conditioning (gamma, mu, resid): %s" % (plan.conditioning(),) if plan.chunks[0] > 1 else ""), flush=True)
                except Exception as e:
                    print("ATTRIBUTION failed:", tag, repr(e), flush=True)
        if k % 3 == 0 and width <= 64 and plan.chunks[0] > 1 and N >= 512 and ok.all():   # (clr_batch_solve: chunked plans, wide ones from N = 512)
            plan.log_likelihood(materialize=True)
            x = plan.solve()
            p = int(rng.randint(B))
            r = ref.RefSolver()
            r.compute(0.0, *coeffs_of(case, p), E, E2, E2, case["t"][p], case["diag"][p])
            want = r.solve(case["y"][p])[:, 0]
            devs["batched solve"] = np.max(np.abs(x[p] - want)) / np.max(np.abs(want))
        if k % 3 == 0 and width <= 64 and ok.all() and N >= 2:
            plan.log_likelihood(materialize=True)
            lz = plan.dot_L(zz)
            p = int(rng.randint(B))
            r = ref.RefSolver()
            r.compute(0.0, *coeffs_of(case, p), E, E2, E2, case["t"][p], case["diag"][p])
            want = r.dot_L(zz[p])[:, 0]
            devs["batched dot_L"] = np.max(np.abs(lz[p] - want)) / np.max(np.abs(want))
            if k % 6 == 0 and B >= 2:
                sp = batch.ShardedBatchedGP(B, N, JR, JC, devices=[0] * min(B, 3))
                try:
                    sp.set_chunks(plan.chunks[0])
                    sp.set_series(case["t"], case["diag"], case["y"])
                    sp.set_coefficients(*coeffs_of(case))
                    sp.materialize()
                    devs["sharded dot_L vs plan"] = np.max(np.abs(sp.dot_L(zz) - lz)) / np.max(np.abs(lz))
                    devs["sharded dot vs plan"] = np.max(np.abs(sp.dot(zz) - kz)) / np.max(np.abs(kz))
                finally:
                    sp.close()
        for name, v in devs.items():
            v = float(v)
            if not (v <= worst.get(name, (0.0,))[0]):
                worst[name] = (v, tag)
            if not (v <= 1e-9):
                bad += 1
                print("ABOVE 1e-9: %s %.2e %s" % (name, v, tag), flush=True)
    finally:
        plan.close()
print("%d cases in %.0f s; worst deviation (value, (B, N, JR, JC, family, chunks)):" % (cases, time.time() - t0))
for name in sorted(worst):
    print("  %-14s %.2e  %s" % (name, worst[name][0], worst[name][1]))
print("cases with a finding:", bad)
