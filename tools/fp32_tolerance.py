# -*- coding: utf-8 -*-
"""BASELINE config 5 asks for the fp32-vs-fp64 tolerance of the width-32 recurrence: this emulates the
reference recurrence (cholesky.h:126-179 fused with :348-357, state before sample n as in DESIGN.md section 2) with the
state S, f and all per-step arithmetic in float32 (features evaluated in fp64 and rounded) against float64, in NumPy."""
import numpy as np, time
def run(dtype, N, JC, seed=3):
    rng = np.random.RandomState(seed)
    t = np.sort(rng.rand(N)); diag = rng.uniform(0.1, 0.2, N) ** 2; y = np.sin(t)
    ac = np.exp(0.1 + 0.1 * rng.randn(JC)); cc = np.exp(2.0 + 0.1 * rng.randn(JC)); dc = np.exp(rng.uniform(0, 3, JC))
    W = 2 * JC
    a = np.zeros(W); a[0::2] = ac; a[1::2] = ac     # u = (a cos, a sin) with b = 0
    c = np.repeat(cc, 2); d = np.repeat(dc, 2)
    S = np.zeros((W, W), dtype); f = np.zeros(W, dtype)
    asum = dtype(ac.sum())
    ld = 0.0; quad = 0.0
    ph = np.zeros(W)
    for n in range(N):
        ang = d * t[n]
        trig = np.where(np.arange(W) % 2 == 0, np.cos(ang), np.sin(ang))      # features in fp64, cast
        u = (a * trig).astype(dtype); v = trig.astype(dtype)
        q = S @ u
        D = dtype(diag[n]) + asum - u @ q
        z = v - q; w = z / D
        x = dtype(y[n]) - u @ f
        ld += np.log(np.float64(D)); quad += np.float64(x) ** 2 / np.float64(D)
        if n + 1 < N:
            phi = np.exp(-c * (t[n + 1] - t[n])).astype(dtype)
            S = (phi[:, None] * phi[None, :]) * (S + np.outer(z, w))
            f = phi * (f + w * x)
    return ld, quad
for N in (2000, 20000):
    t0 = time.time(); r64 = run(np.float64, N, 16); r32 = run(np.float32, N, 16)
    print("N=%d width 32: fp32 vs fp64 rel err: logdet %.2e quad %.2e  (%.1fs)" % (N, abs(r32[0]-r64[0])/abs(r64[0]), abs(r32[1]-r64[1])/abs(r64[1]), time.time()-t0))
