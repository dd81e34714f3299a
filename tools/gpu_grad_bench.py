"""grad_log_likelihood at the headline shape (B = 1024, N = 1e5, width 8 = 2 real + 3 complex, 17 partials): the
plan gradient parallel in n (clr_batch_grad) against the sequential tangent kernel (one wave per (problem, partial))
on a slice of the batch, and one long series through both."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bench import make_inputs, make_inputs_accuracy
from celerite_amd import batch

def seq(coeffs, t, diag, y):
    os.environ["CLR_GRAD_SEQUENTIAL"] = "1"
    try:
        t0 = time.perf_counter()
        out = batch.batch_grad_log_likelihood(*coeffs, t, diag, y)
        return out, time.perf_counter() - t0
    finally:
        del os.environ["CLR_GRAD_SEQUENTIAL"]

for name, maker in (("bench family", make_inputs), ("accuracy family", make_inputs_accuracy)):
    B, N = 1024, 100000
    coeffs, t, diag, y = maker(B, N, 2, 3, 42)
    plan = batch.BatchedGP(B, N, 2, 3)
    plan.set_series(t, diag, y); plan.set_coefficients(*coeffs)
    res = {}
    for mode in ("forward", "reverse-direct-riders", "reverse"):
        plan.set_grad_mode(mode)
        v, g, st = plan.grad_log_likelihood()
        batch.device_synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            v, g, st = plan.grad_log_likelihood()
        dt = (time.perf_counter() - t0) / 3
        res[mode] = g
        print("%s: plan gradient (%s mode) B=%d N=%d: %.2f ms per call (%.1f us per problem), chunks %s, fallbacks %d, ok %d, %s" % (
            name, mode, B, N, dt * 1e3, dt / B * 1e6, plan.chunks, plan.grad_fallbacks(), int((st == 0).sum()), plan.grad_info()), flush=True)
    gs_ = np.max(np.abs(res["forward"]), axis=1, keepdims=True)
    print("   reverse vs forward: %.1e of the largest partial" % np.max(np.abs(res["reverse"] - res["forward"]) / gs_), flush=True)
    plan.close()
    S = 64
    (vs, gs, sts), dts = seq([c[:S] for c in coeffs], t[:S], diag[:S], y[:S])
    (vs, gs, sts), dts = seq([c[:S] for c in coeffs], t[:S], diag[:S], y[:S])
    scale = np.maximum(np.abs(gs), 1e-6 * np.max(np.abs(gs), axis=1, keepdims=True))
    print("   sequential kernel on %d problems: %.1f ms incl. upload (%.2f ms per problem); plan vs sequential: value %.1e grad %.1e" % (
        S, dts * 1e3, dts * 1e3 / S, np.max(np.abs(v[:S] - vs) / np.abs(vs)), np.max(np.abs(g[:S] - gs) / scale)), flush=True)
    # one long series: the object API (a one-problem plan kept inside the solver)
    import celerite_amd
    sol = celerite_amd.CholeskySolver()
    e, e2 = np.empty(0), np.empty((0, 0))
    args = (0.0,) + tuple(c[0] for c in coeffs) + (e, e2, e2, t[0], y[0], diag[0])
    sol.grad_log_likelihood(*args)
    t0 = time.perf_counter()
    for _ in range(5):
        vo, go = sol.grad_log_likelihood(*args)
    do = (time.perf_counter() - t0) / 5
    print("   CholeskySolver.grad_log_likelihood, N=%d: %.2f ms per call" % (N, do * 1e3), flush=True)
    one = [c[:1] for c in coeffs]
    t0 = time.perf_counter()
    for _ in range(3):
        v1, g1, st1 = batch.batch_grad_log_likelihood(*one, t[:1], diag[:1], y[:1])
    d1 = (time.perf_counter() - t0) / 3
    (v2, g2, st2), d2 = seq(one, t[:1], diag[:1], y[:1])
    print("   one series N=%d: one-shot parallel in n %.2f ms (incl. plan set-up and upload), sequential %.1f ms, grad diff %.1e" % (
        N, d1 * 1e3, d2 * 1e3, np.max(np.abs(g1 - g2) / np.maximum(np.abs(g2), 1e-6 * np.max(np.abs(g2))))), flush=True)
