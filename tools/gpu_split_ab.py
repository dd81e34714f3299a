"""A/B of the summarize kernels at the headline shape: single wave (mode 0) vs the role
split with one / two A columns in the trajectory wave (modes 1, 2)."""
import sys, os, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_inputs
from celerite_amd import batch

B, N = int(os.environ.get("B", 1024)), int(os.environ.get("N", 100000))
shapes = [(2, 3)] if len(sys.argv) < 2 else [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]]
for JR, JC in shapes:
    coeffs, t, diag, y = make_inputs(B, N, JR, JC, seed=42)
    plan = batch.BatchedGP(B, N, JR, JC)
    plan.set_series(t, diag, y)
    plan.set_coefficients(*coeffs)
    ref = None
    for mode in (0, 1, 2, 0, 2):
        plan.set_summarize_mode(mode)
        plan.enqueue(); plan.synchronize()
        tot, k = plan.run_timed(10)
        ll, ld, q, st = plan.results()
        if ref is None:
            ref = (ld.copy(), q.copy(), st.copy())
        ok = st == 0
        e1 = float(np.max(np.abs(ld[ok] - ref[0][ok]) / np.abs(ref[0][ok])))
        e2 = float(np.max(np.abs(q[ok] - ref[1][ok]) / np.abs(ref[1][ok])))
        print(json.dumps({"shape": [JR, JC], "mode": mode, "ms_per_step": tot / 10,
                          "summarize_ms": k["summarize"] / 10, "prefix_ms": k["prefix"] / 10,
                          "vs_mode0_logdet": e1, "vs_mode0_quad": e2,
                          "status_equal": bool((st == ref[2]).all()), "replayed": plan.exact_count()}))
    plan.close()
