# -*- coding: utf-8 -*-
"""Calibration of the routing criterion on the CPU (test infrastructure: the host instantiation of csrc/clr_core.h,
tests/hostcheck, with the routing switched off) against the oracle: for every positive definite problem of the
adversarial family the chunk-summary (route 0) result, its deviation from the oracle and the conditioning record
(gamma_max, mu_min, and the measured accuracy eG of G = (I + P Jm)^-1 P).  Usage: narrow|wide <trials>."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_hostcheck as th  # noqa: E402
from _cases import adversarial, coeffs_of  # noqa: E402
from oracle import ref  # noqa: E402

so = "/tmp/libhostcheck_wide.so"
subprocess.check_call(["g++", "-O2", "-std=c++14", "-shared", "-fPIC", "-ffp-contract=off", "-mfma", "-DHOSTCHECK_WIDE",
                       "-o", so, os.path.join(th.HERE, "hostcheck.cpp")])
lib = C.CDLL(so)
dp = C.POINTER(C.c_double)
lib.hostcheck_set_never_replay(1)
mode, trials = sys.argv[1], int(sys.argv[2])
rows = []
for trial in range(trials):
    if mode == "wide":
        JR, JC = 0, 16; N = (200, 1000, 3000)[trial % 3]; chunks = (max(2, N // 100), 16)
    else:
        JR, JC = th.SHAPES[trial % len(th.SHAPES)]
        N = (50, 200, 1000, 3000, 20000)[trial % 5]; chunks = (max(2, N // 40), max(2, N // 8))
    case = adversarial(4, N, JR, JC, seed=9000 + trial)
    l0, d0, q0, s0 = ref.batch_log_likelihood(0.0, *coeffs_of(case), case["t"], case["diag"], case["y"])
    for nchunk in chunks:
        ll, ld, q, st, _ = th.run(lib, JR, JC, nchunk, case, 1)
        flagged = th.run.used_exact.copy()   # certificate / per-chunk error estimate: route 2 whatever the record says
        diag = np.zeros((4, 3)); lib.hostcheck_get_diag(4, diag.ctypes.data_as(dp))
        for p in range(4):
            if s0[p] != 0 or not np.isfinite(d0[p]) or not np.isfinite(q0[p]) or flagged[p]:
                continue
            dev = max(abs(ld[p] - d0[p]) / abs(d0[p]), abs(q[p] - q0[p]) / abs(q0[p]))
            rows.append((diag[p, 0], diag[p, 1], diag[p, 2], dev))
r = np.array(rows)
gam, mu, eg, dev = r.T
print("# %s: %d adversarial problem x chunking combinations the certificate let through (N <= %d)" % (mode, len(r), 3000 if mode == "wide" else 20000))
lc = lambda a: np.corrcoef(np.log10(np.maximum(a, 1e-300)), np.log10(np.maximum(dev, 1e-18)))[0, 1]
print("log-log correlation of the deviation with: gamma %.2f   1/mu %.2f   gamma/mu %.2f   measured eG %.2f   gamma*eG %.2f"
      % (lc(gam), lc(1 / mu), lc(gam / mu), lc(eg), lc(gam * eg)))
def rep(name, m):
    print("%-36s settled from the summaries %4d of %d   worst deviation %.2e   > 1e-11: %d   > 1e-10: %d"
          % (name, m.sum(), len(m), dev[m].max() if m.any() else 0, (dev[m] > 1e-11).sum(), (dev[m] > 1e-10).sum()))
rep("round 2: gamma/mu < 1e6", gam / mu < 1e6)
rep("round 3: gamma < 1e4 & gamma/mu < 1e7", (gam < 1e4) & (gam / mu < 1e7))
rep("round 3, all three: ... & gamma*eG < 3e-9", (gam < 1e4) & (gam / mu < 1e7) & (gam * eg < 3e-9))
rep("         ... & gamma*eG < 1e-9", (gam < 1e4) & (gam / mu < 1e7) & (gam * eg < 1e-9))
rep("         ... & gamma*eG < 1e-8", (gam < 1e4) & (gam / mu < 1e7) & (gam * eg < 1e-8))
rep("         gamma < 2e4 & gamma/mu < 1e7", (gam < 2e4) & (gam / mu < 1e7))
rep("         gamma < 1e4 & gamma/mu < 3e7", (gam < 1e4) & (gam / mu < 3e7))
rep("         gamma * eG < 1e-9", gam * eg < 1e-9)
for lo, hi in [(0, 1e2), (1e2, 1e3), (1e3, 1e4), (1e4, 1e5), (1e5, 1e6), (1e6, 1e300)]:
    m = (gam >= lo) & (gam < hi)
    if m.any():
        print("gamma in [%.0e, %.0e): n=%4d  worst deviation %.2e  median %.2e  dev / (gamma^2 eps) max %.2g"
              % (lo, hi, m.sum(), dev[m].max(), np.median(dev[m]), (dev[m] / (gam[m] ** 2 * 2.2e-16)).max()))
