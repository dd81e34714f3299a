# -*- coding: utf-8 -*-
"""Randomised cross-check of the object API above width 32 against the CPU oracle: random (J_real, J_comp, N), with and
without the hinted right-hand side, every consumer of the factor (dot_solve, solve with 1..3 columns, dot_L, dot,
predict).  Prints the worst deviation per call and every case above 1e-9.  Usage: gpu_fuzz_object_api.py [cases] [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import celerite_amd
from oracle import ref
from _cases import synthetic, coeffs_of, NO_GENERAL

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
worst = {}
bad = 0
t0 = time.time()
for k in range(cases):
    width = int(rng.choice([33, 40, 64, 65, 66, 96, 127, 128, 129, 130, 200, 255, 256, 257, 300, 511, 513, 700]))
    JC = int(rng.randint(0, width // 2 + 1))
    JR = width - 2 * JC
    N = int(rng.choice([1, 2, 3, 5, 63, 64, 65, 200, 257, 1000, 2047, 2048, 2049, 4095, 4096, 4097, 6000, 9000]))
    if width > 300 and N > 4200:
        N = 4100
    case = synthetic(1, max(N, 2), JR, JC, "accuracy", seed=1000 + k)
    cs = list(coeffs_of(case, 0))
    t, diag, y = case["t"][0][:N], case["diag"][0][:N] + 0.05, case["y"][0][:N]
    r = ref.RefSolver()
    r.compute(0.1, *cs, *NO_GENERAL, t, diag)
    s = celerite_amd.CholeskySolver()
    hint = bool(rng.randint(2))
    if hint:
        s._hint_rhs(y)
    s.compute(0.1, *cs, *NO_GENERAL, t, diag)
    nrhs = int(rng.randint(1, 4))
    b = rng.randn(N, nrhs)
    devs = {}
    devs["logdet"] = abs(s.log_determinant() - r.log_determinant()) / max(abs(r.log_determinant()), 1e-300)
    devs["dot_solve(y)"] = abs(s.dot_solve(y) - r.dot_solve(y)) / abs(r.dot_solve(y))
    devs["dot_solve"] = abs(s.dot_solve(b[:, 0]) - r.dot_solve(b[:, 0])) / abs(r.dot_solve(b[:, 0]))
    want = r.solve(b)
    devs["solve"] = np.max(np.abs(s.solve(b) - want)) / np.max(np.abs(want))
    want = r.dot_L(b)
    devs["dot_L"] = np.max(np.abs(s.dot_L(b) - want)) / np.max(np.abs(want))
    want = r.dot(0.1, *cs, *NO_GENERAL, t, b)
    devs["dot"] = np.max(np.abs(celerite_amd.CholeskySolver().dot(0.1, *cs, *NO_GENERAL, t, b) - want)) / np.max(np.abs(want))
    if N >= 2:
        xs = np.sort(rng.uniform(t.min() - 0.1, t.max() + 0.1, 17))
        want = r.predict(y, xs)
        devs["predict"] = np.max(np.abs(s.predict(y, xs) - want)) / max(np.max(np.abs(want)), 1e-300)
    for name, v in devs.items():
        v = float(v)
        if not (v <= worst.get(name, (0.0,))[0]):
            worst[name] = (v, (JR, JC, N, nrhs, hint))
        if not (v <= 1e-9):
            bad += 1
            print("ABOVE 1e-9: %s %.2e  JR %d JC %d N %d nrhs %d hint %s" % (name, v, JR, JC, N, nrhs, hint), flush=True)
print("%d cases in %.0f s; worst deviation per call (value, (JR, JC, N, nrhs, hinted)):" % (cases, time.time() - t0))
for name in sorted(worst):
    print("  %-14s %.2e  %s" % (name, worst[name][0], worst[name][1]))
print("cases above 1e-9:", bad)
