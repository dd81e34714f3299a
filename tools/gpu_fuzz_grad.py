"""Randomised cross-check of CholeskySolver.grad_log_likelihood parallel in n (widths 1..64: reverse mode at widths
1..8, chunk-wise tangents at the padded widths 16 / 32 / 64) against the sequential tangent kernel (CLR_GRAD_SEQUENTIAL)
on random (J_real, J_comp, N), dense and sparse sampling, zero and non-zero jitter, an occasional indefinite kernel.
Usage: gpu_fuzz_grad.py [cases] [seed]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import celerite_amd  # noqa: E402
from celerite_amd import batch  # noqa: E402

NO_GENERAL = (np.empty(0), np.empty((0, 0)), np.empty((0, 0)))
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
worst = {}
bad = 0
t0 = time.time()
s = celerite_amd.CholeskySolver()
for k in range(cases):
    width = int(rng.choice([1, 3, 5, 8, 9, 12, 16, 17, 25, 32, 33, 34, 40, 47, 48, 56, 63, 64]))
    JC = int(rng.randint(0, width // 2 + 1))
    JR = width - 2 * JC
    N = int(rng.choice([1024, 2048, 4096, 4097, 5000, 8192, 12000, 30000]))
    dense = rng.randint(2) == 0
    x = np.sort(rng.uniform(0, (1.0 if dense else 0.05 * N), N))
    diag = rng.uniform(0.1, 0.3, N)
    y = rng.randn(N)
    co = [np.exp(rng.uniform(-1, 1, JR)), np.exp(rng.uniform(-2, 0, JR)), np.exp(rng.uniform(-1, 0, JC)),
          0.2 * rng.rand(JC), np.exp(rng.uniform(-2, 0, JC)), np.exp(rng.uniform(-1, 1.5, JC))]
    jitter = 0.0 if rng.randint(3) == 0 else float(rng.uniform(0.01, 0.5))
    indefinite = rng.randint(8) == 0
    if indefinite:
        co[0 if JR else 2] = -80.0 * co[0 if JR else 2]
    args = (jitter,) + tuple(co) + NO_GENERAL + (x, y, diag)
    tag = (JR, JC, N, "dense" if dense else "sparse", jitter > 0)
    res = {}
    for mode in ("par", "seq"):
        try:
            if mode == "seq":
                with batch.option("CLR_GRAD_SEQUENTIAL"):
                    res[mode] = s.grad_log_likelihood(*args)
            else:
                res[mode] = s.grad_log_likelihood(*args)
        except celerite_amd.solver.LinAlgError:
            res[mode] = None
    if (res["par"] is None) != (res["seq"] is None):
        bad += 1
        print("STATUS differs:", tag, res["par"] is None, res["seq"] is None, flush=True)
        continue
    if res["par"] is None:
        continue
    (v, g), (v1, g1) = res["par"], res["seq"]
    devs = {"value": abs(v - v1) / abs(v1), "partials (of the largest)": np.max(np.abs(g - g1)) / np.max(np.abs(g1))}
    if (g[0] == 0.0) != (jitter == 0.0):
        bad += 1
        print("zero-jitter rule:", tag, g[0], flush=True)
    for name, val in devs.items():
        val = float(val)
        if not (val <= worst.get(name, (0.0,))[0]):
            worst[name] = (val, tag)
        if not (val <= 1e-9):
            bad += 1
            print("ABOVE 1e-9: %s %.2e %s" % (name, val, tag), flush=True)
print("%d cases in %.0f s; worst deviation parallel in n vs sequential tangent kernel (value, (JR, JC, N, sampling, jitter)):" % (cases, time.time() - t0))
for name in sorted(worst):
    print("  %-28s %.2e  %s" % (name, worst[name][0], worst[name][1]))
print("cases with a finding:", bad)
