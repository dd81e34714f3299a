# -*- coding: utf-8 -*-
"""Per-problem deviations of the replay-free route on tests/_cases.adversarial, against the oracle
(and, for forced exact runs, of the replay route): the data the certificate thresholds are set by."""
import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from celerite_amd import batch
from oracle import ref
from _cases import adversarial, coeffs_of, ALL_WIDTH_SHAPES

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 300
WIDE = bool(os.environ.get("WIDE"))   # widths 9..32 (the wave-per-chunk scan) instead of 1..8
WIDE_SHAPES = [(9, 0), (1, 4), (0, 8), (16, 0), (0, 16), (4, 14), (2, 7), (10, 11)]
rows = []
mismatch = 0
for trial in range(trials):
    JR, JC = (WIDE_SHAPES[trial % len(WIDE_SHAPES)] if WIDE else ALL_WIDTH_SHAPES[trial % len(ALL_WIDTH_SHAPES)])
    N = ((600, 2000, 6000)[trial % 3] if WIDE else (50, 200, 1000, 3000, 20000)[trial % 5])
    case = adversarial(4, N, JR, JC, seed=5000 + trial)
    l0, d0, q0, s0 = ref.batch_log_likelihood(0.0, *coeffs_of(case), case["t"], case["diag"], case["y"])
    plan = batch.BatchedGP(4, N, JR, JC)
    plan.set_series(case["t"], case["diag"], case["y"])
    plan.set_coefficients(*coeffs_of(case))
    if os.environ.get("CERT") is not None:
        cert = float(os.environ["CERT"])   # CERT=0: routing by the conditioning record switched off
        plan.set_certificate(cert, float(os.environ.get("RESID", "1e-12")), max_gamma=float(os.environ.get("GAMMA", "0" if cert == 0 else "1e4")),
                             max_gamma_error=float(os.environ.get("GAMMAEG", "0" if cert == 0 else "3e-9")))
    for nchunk in ((4, 9) if WIDE else (max(2, N // 40), max(2, N // 8))):
        plan.set_chunks(nchunk)
        ll, ld, q, st = plan.log_likelihood()
        mismatch += int((st != s0).sum())
        fl = plan.exact_levels()
        gam, mu = plan.conditioning()
        plan.set_exact(True)
        ll2, ld2, q2, st2 = plan.log_likelihood()
        fl2 = plan.exact_levels()
        plan.conditioning(); res2 = plan.last_residual
        plan.set_exact(False)
        for p in range(4):
            if s0[p] != 0 or not np.isfinite(d0[p]) or not np.isfinite(q0[p]):
                continue
            rows.append(dict(trial=trial, JR=JR, JC=JC, N=N, nchunk=nchunk, p=p, replayed=bool(fl[p]), level=int(fl[p]), level2=int(fl2[p]), resid=float(res2[p]), gamma=float(gam[p]), mu=float(mu[p]),
                             st=int(st[p]), st2=int(st2[p]),
                             eld=abs(ld[p] - d0[p]) / abs(d0[p]), eq=abs(q[p] - q0[p]) / abs(q0[p]),
                             eld2=abs(ld2[p] - d0[p]) / abs(d0[p]), eq2=abs(q2[p] - q0[p]) / abs(q0[p]),
                             ld=float(d0[p]), q=float(q0[p])))
    plan.close()
print("status mismatches against the oracle: %d" % mismatch)
if os.environ.get("DUMP"):
    json.dump(rows, open(os.environ["DUMP"], "w"))
free = [r for r in rows if not r["replayed"]]
rep = [r for r in rows if r["replayed"]]
def stats(name, xs):
    xs = np.array(xs) if len(xs) else np.zeros(1)
    print("%-34s n=%5d  max %.2e  p99 %.2e  p90 %.2e  >1e-10: %d  >1e-12: %d" % (
        name, len(xs), xs.max(), np.percentile(xs, 99), np.percentile(xs, 90), (xs > 1e-10).sum(), (xs > 1e-12).sum()))
print("problems (status ok in the oracle): %d, settled replay-free %d, replayed %d" % (len(rows), len(free), len(rep)))
stats("replay-free  logdet rel", [r["eld"] for r in free])
stats("replay-free  quad   rel", [r["eq"] for r in free])
stats("flagged->replay logdet rel", [r["eld"] for r in rep])
stats("flagged->replay quad   rel", [r["eq"] for r in rep])
stats("forced exact, all: logdet rel", [r["eld2"] for r in rows])
stats("forced exact, all: quad   rel", [r["eq2"] for r in rows])
worst = sorted(free, key=lambda r: -max(r["eld"], r["eq"]))[:8]
for r in worst:
    print(json.dumps(r))
worst = sorted(rows, key=lambda r: -max(r["eld2"], r["eq2"]))[:8]
print("worst of the forced-exact (replay from scanned starts):")
for r in worst:
    print(json.dumps(r))

# how the replay-free deviations relate to the conditioning record
gg = np.array([r["gamma"] for r in free]); ee = np.array([max(r["eld"], r["eq"]) for r in free])
for lo, hi in [(0, 1e2), (1e2, 1e3), (1e3, 1e4), (1e4, 1e5), (1e5, 1e6), (1e6, 1e300)]:
    m = (gg >= lo) & (gg < hi)
    if m.any():
        print("gamma in [%.0e, %.0e): n=%4d  worst deviation %.2e  median %.2e" % (lo, hi, m.sum(), ee[m].max(), np.median(ee[m])))
g = np.array([r["gamma"] / max(r["mu"], 1e-300) for r in free])
e = np.array([max(r["eld"], r["eq"]) for r in free])
for lo, hi in [(0, 1e3), (1e3, 1e4), (1e4, 1e5), (1e5, 1e6), (1e6, 1e8), (1e8, 1e300)]:
    m = (g >= lo) & (g < hi)
    if m.any():
        print("gamma/mu in [%.0e, %.0e): n=%4d  worst deviation %.2e  median %.2e" % (lo, hi, m.sum(), e[m].max(), np.median(e[m])))
from _cases import synthetic
for fam in ("bench", "accuracy"):
    for (JR, JC) in [(2, 3), (0, 4), (8, 0), (5, 0), (6, 1), (3, 0), (1, 1), (0, 2)]:
        for N, nchunk in [(3000, 64), (100000, 64), (20000, 256)]:
            case = synthetic(8, N, JR, JC, fam, seed=3)
            plan = batch.BatchedGP(8, N, JR, JC); plan.set_chunks(nchunk)
            plan.set_series(case["t"], case["diag"], case["y"]); plan.set_coefficients(*coeffs_of(case))
            plan.set_certificate(0.0, 1e-12, max_gamma=0.0, max_gamma_error=0.0)
            ll, ld, q, st = plan.log_likelihood(); gam, mu = plan.conditioning(); plan.close()
            S = 2 if N > 50000 else 8
            l0, d0, q0, s0 = ref.batch_log_likelihood(0.0, *[c[:S] for c in coeffs_of(case)], case["t"][:S], case["diag"][:S], case["y"][:S])
            dev = max(np.max(np.abs(ld[:S] - d0) / np.abs(d0)), np.max(np.abs(q[:S] - q0) / np.abs(q0)))
            print("%-8s (%d,%d) N=%6d chunks %3d: gamma_max %.3g  mu_min %.3g  gamma/mu %.3g  deviation %.2e" % (fam, JR, JC, N, nchunk, gam.max(), mu.min(), (gam / mu).max(), dev))
