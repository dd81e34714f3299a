# -*- coding: utf-8 -*-
"""Headline benchmark: GP log-likelihoods / second on MI355X.

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[2] per GPU; configs[3] = the same on 8 GPUs):
    B = 1024 independent (series x hyper-parameter draw) problems per GPU,
    N = 100000 samples each, width J = 8 (2 real + 3 complex celerite terms,
    the examples/benchmark/run.py:80-84 recipe with 10 % log-parameter scatter),
    fp64, synthetic "bench" family (run.py:66-69): t = sort(U(0,1)), sigma =
    U(0.1,0.2), y = sin t.

One "step" = what an optimiser / MCMC iteration pays (celerite.py:160-219 per
problem): a FRESH draw of hyper-parameters for all B problems goes host -> HBM
(`set_coefficients`), the five kernels run (summarize -> prefix -> correct ->
replay -> finalize, csrc/), and the B log-likelihoods come back to the host
(`results`, which synchronises).  The series are resident in HBM (uploaded
before the timed region, in the public API's row-major layout).  `value` is
that loop's rate; `device_only` is the same kernels launched back to back
without the per-step transfers (what round 1 reported as `value`).

Per-kernel times come from HIP events recorded on the plan's stream around
every kernel INSIDE the timed loop (`clr_batch_set_profiling`).

Multi-GPU: the batch axis shards embarrassingly -- one process per GPU, no
collective on the data path; torch.distributed (gloo) is used
only for the rendezvous, the timing barrier and the max-over-ranks reduction.
Weak scaling: every rank evaluates its own B problems.  Launched without
WORLD_SIZE and with --gpus N > 1, this script starts its own N ranks
(torch.distributed.run) and fails if fewer than N GPUs are visible.  (The
single-process alternative is the product's `ShardedBatchedGP` /
`clr_sharded_*`: one host thread per GPU.)

Rank 0 prints ONE JSON line on stdout, the LAST thing written there and below 6 KB (the driver keeps the
last 8 KB of stdout): the contract's keys, `config`, `roofline` (the dominant kernel + the promoted numbers of
the side legs) and `cpu_baseline` -- `headline_line()`, held to that size by tests/test_host_api.py.  The
COMPLETE record goes to `gpurun_out/bench_full.json` (or $CLR_BENCH_FULL) and to stderr: `configs` in it
carries BASELINE configs 0, 1 and 4 (object API at N = 1e3; B = 256 x N = 1e4 width 4; B = 256 x N = 1e5
width 32), each with its own time, rate, roofline and parity on a sample (and, for the batch configs, the
histogram of problems by route and the conditioning record).  `accuracy_family` is SURVEY.md 8(d)'s second
input family at the headline shape (sparse sampling: the plan runs the warm-started plain recurrence instead
of the scan); `sharded_product_path` times the product's own sharding (one process, one host thread + plan
per shard: batch.ShardedBatchedGP) next to the process-per-GPU number `--gpus N` reports; `value_steady` is
the 2.5 s steady-state leg of the real loop.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time
import timeit

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# MI355X datasheet: 256 CU x 4 SIMD x 16 FMA lanes x 2 flop x 2.4 GHz (matrix = vector rate for fp64)
PEAK_FP64_TFLOPS = 78.6
# what the VECTOR ALU sustains on an all-FMA stream with every SIMD busy: one wave64 fp64 FMA per
# 2.46 ns per SIMD (tools/microbench/issue_rates2.hip, profiles/r02a_issue_rates2.txt)
MEASURED_VALU_FMA_TFLOPS = 1024 * 64 * 2 / 2.46e-9 / 1e12
PEAK_HBM_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md (spec; ~6300 measured)
# ... and what THIS box sustains, measured live by main() (clr_device_measure_fp64: rate, shader clock, SIMD cycles per FMA)
FP64_LIVE = {}


def csrc_fingerprint():
    """sha1 over the kernel sources (celerite_amd/csrc/*.h, *.hip without the api_* files): profiles/pmc_latest.json carries the
    fingerprint of the build its counters were taken from; a different one means the committed HBM-traffic figures
    describe OTHER kernels and `roofline.traffic` is nulled until someone re-profiles."""
    import glob
    import hashlib
    h = hashlib.sha1()
    for path in sorted(glob.glob(os.path.join(ROOT, "celerite_amd", "csrc", "*"))):
        # (the DEVICE code: the api_* translation units, the host-only .cpp files and the pybind module do not change what
        #  a kernel moves; the chunk count the counters were taken at is compared separately)
        if path.endswith((".h", ".hip")) and not os.path.basename(path).startswith("api_"):
            h.update(os.path.basename(path).encode())
            h.update(open(path, "rb").read())
    return h.hexdigest()[:16]


def pmc_record():
    """profiles/pmc_latest.json if its counters belong to this build, else None."""
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", "pmc_latest.json")))
    except Exception:
        return None
    return rec if rec.get("csrc_fingerprint") == csrc_fingerprint() else None


def shard_bounds(total, rank, world):
    """Contiguous slice [lo, hi) of a `total`-long batch axis owned by `rank`."""
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def algorithmic_flops_per_loglik(N, W):
    """F_alg = N (5.5 W^2 + 9.5 W + 3) fp64 flop (fma = 2)."""
    return N * (5.5 * W * W + 9.5 * W + 3.0)


def algorithmic_bytes_per_loglik(N):
    """t, diag, y in + 16 B out."""
    return 24.0 * N + 16.0


def pmc_traffic(kernel, B, N, JR, JC, chunks):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes
    (profiles/pmc_latest.json: FETCH_SIZE x 2 + WRITE_SIZE, see the file), if they
    were taken at this configuration; None otherwise (PMC cannot be read live)."""
    try:
        rec = pmc_record()   # (None when the kernels changed since the counters were collected)
        c = rec["config"]
        if (c["batch"], c["N"], c["J_real"], c["J_comp"], c["chunks"]) == (B, N, JR, JC, chunks):
            return rec["traffic_bytes_per_launch"].get(kernel)
    except Exception:
        pass
    return None


def pmc_traffic_other(key_prefix):
    """HBM bytes per launch of the dominant kernel of one of the other shapes (profiles/pmc_latest.json:
    other_shapes), by the prefix of its label; None if it was not measured."""
    try:
        rec = pmc_record()
        for k, v in rec.get("other_shapes", {}).items():
            if k.startswith(key_prefix):
                return v
    except Exception:
        pass
    return None


def make_inputs(B, N, J_real, J_comp, seed, d_spread=False):
    rng = np.random.RandomState(seed)
    t = np.sort(rng.rand(B, N), axis=1)
    sig = rng.uniform(0.1, 0.2, (B, N))
    y = np.sin(t)
    a_real = np.exp(1.0 + 0.1 * rng.randn(B, J_real))
    c_real = np.exp(0.1 + 0.1 * rng.randn(B, J_real))
    a_comp = np.exp(0.1 + 0.1 * rng.randn(B, J_comp))
    b_comp = np.zeros((B, J_comp))
    c_comp = np.exp(2.0 + 0.1 * rng.randn(B, J_comp))
    if d_spread:  # log d spread over U(0, 3)
        d_comp = np.exp(rng.uniform(0.0, 3.0, (B, J_comp)))
    else:
        d_comp = np.exp(1.6 + 0.1 * rng.randn(B, J_comp))
    return (a_real, c_real, a_comp, b_comp, c_comp, d_comp), t, sig ** 2, y


def make_inputs_accuracy(B, N, J_real, J_comp, seed):
    """The paper's accuracy family (paper/figures/error/error.py:24-25): t = sort(U(0, 0.8 N)) -- mean spacing 0.8 --
    sigma = U(1, 1.5), y = N(0, 1); coefficients as make_inputs."""
    rng = np.random.RandomState(seed)
    t = np.sort(rng.uniform(0, 0.8 * N, (B, N)), axis=1)
    sig = rng.uniform(1.0, 1.5, (B, N))
    y = rng.randn(B, N)
    a_real = np.exp(1.0 + 0.1 * rng.randn(B, J_real))
    c_real = np.exp(0.1 + 0.1 * rng.randn(B, J_real))
    a_comp = np.exp(0.1 + 0.1 * rng.randn(B, J_comp))
    b_comp = np.zeros((B, J_comp))
    c_comp = np.exp(2.0 + 0.1 * rng.randn(B, J_comp))
    d_comp = np.exp(1.6 + 0.1 * rng.randn(B, J_comp))
    return (a_real, c_real, a_comp, b_comp, c_comp, d_comp), t, sig ** 2, y


def fresh_draws(coeffs, count, seed):
    """`count` hyper-parameter proposals around `coeffs` (1 % log-normal steps, b_comp kept):
    what an MCMC sampler hands over per iteration."""
    rng = np.random.RandomState(seed)
    out = []
    for _ in range(count):
        out.append(tuple(np.ascontiguousarray(c * np.exp(0.01 * rng.randn(*c.shape))) for c in coeffs))
    return out


def best_of_3(fn, min_time=0.2):
    """celerite/timer.py:8-15: double the repeat count until three repeats take >= 0.2 s in
    total, return the best per-call time."""
    k = 1
    while True:
        times = timeit.repeat(fn, repeat=3, number=k)
        if sum(times) >= min_time or k >= 1 << 20:
            return min(times) / k
        k *= 2


class Dist(object):
    """Rendezvous + barrier + max-reduce; a no-op for a single process."""

    def __init__(self):
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.pg = None
        if self.world > 1:
            import torch.distributed as dist

            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29500")
            dist.init_process_group(backend="gloo", rank=self.rank, world_size=self.world)
            self.pg = dist

    def barrier(self):
        if self.pg:
            self.pg.barrier()

    def max(self, value):
        if not self.pg:
            return float(value)
        import torch

        t = torch.tensor([float(value)], dtype=torch.float64)
        self.pg.all_reduce(t, op=self.pg.ReduceOp.MAX)
        return float(t[0])

    def close(self):
        if self.pg:
            self.pg.destroy_process_group()


def spawn_ranks(n, argv):
    """--gpus n without a launcher: start n ranks of this script on this node."""
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    return subprocess.call(cmd)


def real_loop(plan, draws, steps, offset=0):
    """`steps` optimiser-style evaluations: upload a fresh draw, launch, fetch the results."""
    out = None
    evaluate = getattr(plan, "evaluate", None)
    for k in range(steps):
        d = draws[(offset + k) % len(draws)]
        if evaluate is not None:            # one library call: set_coefficients + enqueue + results (clr_batch_evaluate)
            out = evaluate(*d)
        else:
            plan.set_coefficients(*d)
            plan.enqueue()
            out = plan.results()
    return out


def roofline_block(per_kernel_ms, B, N, W, traffic=None):
    dom = max(per_kernel_ms, key=per_kernel_ms.get)
    dom_s = per_kernel_ms[dom] * 1e-3
    step_s = sum(per_kernel_ms.values()) * 1e-3
    flops = B * algorithmic_flops_per_loglik(N, W)
    bytes_ = B * algorithmic_bytes_per_loglik(N)
    ach = flops / dom_s / 1e12
    return {
        "kernel": dom, "bound": "fp64_valu", "achieved": ach, "peak": PEAK_FP64_TFLOPS, "unit": "TFLOP/s",
        "frac": ach / PEAK_FP64_TFLOPS, "traffic": traffic, "launch_ms": per_kernel_ms[dom],
        "algorithmic_flops_per_launch": flops,
        "frac_of_measured_valu_fma_rate": ach / FP64_LIVE.get("tflops", MEASURED_VALU_FMA_TFLOPS),
        "measured_valu_fma_rate_tflops": FP64_LIVE.get("tflops", MEASURED_VALU_FMA_TFLOPS),
        "measured_valu_fma_rate_note": ("measured on this box before the timed region (clr_device_measure_fp64, two waves per SIMD "
                                        "issuing independent v_fma_f64): %.1f TFLOP/s at a shader clock of %.0f MHz, %.2f SIMD cycles "
                                        "per FMA -- the datasheet's 78.6 is 4 cycles at 2400 MHz: under full fp64 load the clock is "
                                        "the larger part of the gap, the issue rate the rest (profiles/r05c_clock_under_fp64_load.txt)"
                                        % (FP64_LIVE["tflops"], FP64_LIVE["clock_mhz"], FP64_LIVE["cycles_per_fma"])) if FP64_LIVE else
                                       "round 2's microbenchmark (one FMA per 2.46 ns per SIMD)",
        "shader_clock_mhz_under_fp64_load": FP64_LIVE.get("clock_mhz"),
        "simd_cycles_per_fp64_fma": FP64_LIVE.get("cycles_per_fma"),
        "hbm_view": {"bound": "hbm", "achieved": bytes_ / dom_s / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                     "frac": bytes_ / dom_s / 1e9 / PEAK_HBM_GBS, "algorithmic_bytes_per_launch": bytes_},
        "whole_step": {"achieved_tflops": flops / step_s / 1e12,
                       "frac_fp64": flops / step_s / 1e12 / PEAK_FP64_TFLOPS},
    }


def rel_err(a, b):
    return float(np.max(np.abs(a - b) / np.abs(b))) if len(a) else 0.0


def batch_config(name, B, N, JR, JC, steps, sample, seed, d_spread=False):
    """One of the other BASELINE batch configurations: real loop + device-only rate, roofline of
    its dominant kernel, parity against the oracle on `sample` problems, CPU time beside it."""
    from celerite_amd import batch
    from oracle import ref

    W = JR + 2 * JC
    coeffs, t, diag, y = make_inputs(B, N, JR, JC, seed, d_spread=d_spread)
    plan = batch.BatchedGP(B, N, JR, JC)
    try:
        plan.set_series(t, diag, y)
        draws = [coeffs] + fresh_draws(coeffs, 3, seed + 1)
        real_loop(plan, draws, 2)
        batch.device_synchronize()
        t0 = time.perf_counter()
        real_loop(plan, draws, steps, offset=1)     # (no events inside this loop: seven records cost ~35 us per step)
        dt = time.perf_counter() - t0
        plan.set_profiling(True)
        real_loop(plan, draws, steps, offset=1)     # the same loop again with every kernel bracketed
        kms, nrec = plan.profile()
        plan.set_profiling(False)
        per = {k: v / max(nrec, 1) for k, v in kms.items()}
        # routes of every draw of the timed loop (a draw with one problem on the checked route pays a chunk-time of replay:
        # the kernels_ms above average over the draws)
        levels_by_draw = []
        for dr in draws[1:]:
            plan.set_coefficients(*dr)
            plan.log_likelihood()
            levels_by_draw.append([int(v) for v in np.bincount(plan.exact_levels(), minlength=3)[:3]])
        plan.set_coefficients(*coeffs)
        ll, ld, q, st = plan.log_likelihood()
        levels = np.bincount(plan.exact_levels(), minlength=3)[:3]
        gam, mu = plan.conditioning()
        eg = plan.measured_error()
        dev_ms, _ = plan.run_timed(steps, relayout_each_step=False)
        chunks = plan.chunks
        prefix_plan = plan.prefix_plan if W <= 8 else (0, [], [chunks[0]])
        # the batched solve on this plan's materialised factor (round 6: widths 9..64 too), problem 0 against the oracle's solve
        solve = None
        if W == 32:
            try:
                mat_ms, _ = plan.run_timed(2, materialize=True, relayout_each_step=False)
                plan.solve()
                x = plan.solve()
                r = ref.RefSolver()
                e_, e2_ = np.empty(0), np.empty((0, 0))
                r.compute(0.0, *[c[0] for c in coeffs], e_, e2_, e2_, t[0], diag[0])
                x0 = r.solve(y[0])[:, 0]
                solve = {"what": "clr_batch_solve: K^-1 y for all %d problems from the wide plan's materialised factor "
                                 "(wsweep affine scans, grid.z = problem)" % B,
                         "materialising_step_ms": mat_ms / 2, "solve_device_ms": plan.solve_device_ms(),
                         "problem0_vs_oracle_rel": float(np.max(np.abs(x[0] - x0)) / np.max(np.abs(x0)))}
                del x
            except Exception as e:  # a failing side leg must not lose the line
                solve = {"error": repr(e)}
    finally:
        plan.close()
    S = min(sample, B)
    sub = [c[:S] for c in coeffs]
    l0, d0, q0, s0 = ref.batch_log_likelihood(0.0, *sub, t[:S], diag[:S], y[:S], nthreads=1)
    cpu = best_of_3(lambda: ref.batch_log_likelihood(0.0, *[c[:1] for c in coeffs], t[:1], diag[:1], y[:1],
                                                     nthreads=1))
    ok = s0 == 0
    return {
        "workload": name, "batch": B, "N": N, "width": W, "J_real": JR, "J_comp": JC,
        "scan_chunks": chunks[0], "chunk_len": chunks[1], "steps": steps,
        "prefix_plan": {"levels": prefix_plan[0], "groups": prefix_plan[1], "elements_per_level": prefix_plan[2]},
        "levels": {"what": "problems by route: 0 settled from the chunk summaries, 1 checked chunked replay, "
                           "2 sequential recurrence", "histogram": [int(v) for v in levels],
                   "histograms_of_the_other_draws_of_the_timed_loop": levels_by_draw},
        # (the one-launch path of short narrow problems settles them without leaving a per-chunk record: zeros)
        "conditioning": {"gamma_max": float(np.max(gam)), "mu_min": float(np.min(mu)),
                         "gamma_over_mu_max": float(np.max(gam / np.where(mu > 0, mu, np.inf))),
                         "measured_G_error_max": float(np.max(eg)), "gamma_times_error_max": float(np.max(gam * eg))},
        "ms_per_step": dt / steps * 1e3, "value": B * steps / dt, "unit": "log-likelihoods/s",
        "device_only": {"ms_per_step": dev_ms / steps, "value": B / (dev_ms / steps * 1e-3)},
        "kernels_ms": per,
        "roofline": roofline_block(per, B, N, W, pmc_traffic_other("config4") if (B, N, W) == (256, 100000, 32) else None),
        "cpu_oracle": {"ms_per_loglik": cpu * 1e3, "value": 1.0 / cpu, "cores": 1, "timing": "best of 3 (timer.py)"},
        "parity": {"problems_checked": int(S), "status_equal": bool(np.array_equal(st[:S], s0)),
                   "logdet_rel_max": rel_err(ld[:S][ok], d0[ok]), "quad_rel_max": rel_err(q[:S][ok], q0[ok]),
                   "tolerance": 1e-10},
        "batched_solve": solve,
    }


def accuracy_family_block(B, N, JR, JC, steps, sample, seed):
    """SURVEY.md 8(d)'s second input family at the headline shape: sparse sampling, the state is forgotten within tens
    of samples, so the plan runs the warm-started plain recurrence (csrc/clr_batch_kernels.h: warm_kernel) instead of
    the scan.  Real loop, device-only rate, the scan path on the same inputs for comparison, parity on a sample."""
    from celerite_amd import batch
    from oracle import ref

    W = JR + 2 * JC
    coeffs, t, diag, y = make_inputs_accuracy(B, N, JR, JC, seed)
    plan = batch.BatchedGP(B, N, JR, JC)
    try:
        plan.set_series(t, diag, y)
        draws = [coeffs] + fresh_draws(coeffs, 3, seed + 1)
        real_loop(plan, draws, 3)
        plan.set_profiling(2)                        # (events around the dominant kernel only)
        batch.device_synchronize()
        t0 = time.perf_counter()
        real_loop(plan, draws, steps, offset=1)
        dt = time.perf_counter() - t0
        kms, nrec = plan.profile()
        plan.set_profiling(False)
        per = {k: v / max(nrec, 1) for k, v in kms.items()}
        plan.set_coefficients(*coeffs)
        ll, ld, q, st = plan.log_likelihood()
        warm = plan.warm_start()
        dev_ms, dev_k = plan.run_timed(steps, relayout_each_step=False)
        # the scan path on the same inputs (warm start switched off)
        plan.set_warm_start(0)
        plan.set_coefficients(*coeffs)
        ll_s, ld_s, q_s, st_s = plan.log_likelihood()
        scan_kernel = plan.summarize_kernel()
        scan_ms, scan_k = plan.run_timed(max(steps // 2, 1), relayout_each_step=False)
    finally:
        plan.close()
    S = min(sample, B)
    l0, d0, q0, s0 = ref.batch_log_likelihood(0.0, *[c[:S] for c in coeffs], t[:S], diag[:S], y[:S], nthreads=1)
    ok = s0 == 0
    roof = roofline_block({"warm recurrence + boundary check": per["summarize"]}, B, N, W,
                          pmc_traffic_other("accuracy family") if (B, N, W, bool(warm["active"])) == (1024, 100000, 8, True) else None)
    return {
        "workload": "accuracy family (paper/figures/error/error.py:24-25): batch=%d x N=%d, width %d (%d real + %d "
                    "complex), t = sort(U(0, 0.8 N)), sigma = U(1, 1.5), y = N(0, 1)" % (B, N, W, JR, JC),
        "path": "warm-started plain recurrence per chunk + boundary check" if warm["active"] else "scan",
        "warm_start": warm, "steps": steps,
        "ms_per_step": dt / steps * 1e3, "value": B * steps / dt, "unit": "log-likelihoods/s",
        "device_only": {"ms_per_step": dev_ms / steps, "value": B / (dev_ms / steps * 1e-3),
                        "kernels_ms": {k: v / steps for k, v in dev_k.items()}},
        "kernels_ms": per,
        "kernels_note": "slot `summarize` = warm_kernel + warm_check_kernel (no relayout, prefix, correct or replay "
                        "kernel runs on this path)",
        "roofline": roof,
        "scan_path_same_inputs": {"summarize_kernel": scan_kernel, "ms_per_step": scan_ms / max(steps // 2, 1),
                                  "kernels_ms": {k: v / max(steps // 2, 1) for k, v in scan_k.items()},
                                  "vs_warm_logdet_rel": rel_err(ld[st == 0], ld_s[st == 0]),
                                  "vs_warm_quad_rel": rel_err(q[st == 0], q_s[st == 0])},
        "parity": {"problems_checked": int(S), "status_equal": bool(np.array_equal(st[:S], s0)),
                   "logdet_rel_max": rel_err(ld[:S][ok], d0[ok]), "quad_rel_max": rel_err(q[:S][ok], q0[ok]),
                   "tolerance": 1e-10},
        "status_not_ok": int((st != 0).sum()),
    }


def sharded_block(B, N, JR, JC, nshards, steps, seed, tile=1):
    """The PRODUCT's multi-GPU path (clr_sharded_* / batch.ShardedBatchedGP: one process, one host thread + plan per
    shard, no collective): `nshards` shards over the visible GPUs (round robin; shards share a GPU when there are
    fewer GPUs than shards).  Real loop (evaluate = coefficients in, B results out on every shard concurrently).
    tile > 1: B / tile distinct series, each with `tile` hyper-parameter draws (BASELINE configs[3] = 8 x 1024: the
    series arrays are tiled on the host -- generating 8192 sorted series of 1e5 samples would take a minute)."""
    from celerite_amd import batch

    ndev = batch.device_count()
    devices = [s % ndev for s in range(nshards)]
    coeffs, t, diag, y = make_inputs(B // tile, N, JR, JC, seed)
    if tile > 1:
        more = fresh_draws(coeffs, tile - 1, seed + 7)
        coeffs = tuple(np.concatenate([coeffs[i]] + [m[i] for m in more], axis=0) for i in range(6))
        t, diag, y = (np.tile(a, (tile, 1)) for a in (t, diag, y))
    draws = [coeffs] + fresh_draws(coeffs, 3, seed + 1)
    plan = batch.ShardedBatchedGP(B, N, JR, JC, devices=devices)
    try:
        t0 = time.perf_counter()
        plan.set_series(t, diag, y)
        set_series_s = time.perf_counter() - t0
        for k in range(3):
            out = plan.evaluate(*draws[k % len(draws)])
        t0 = time.perf_counter()
        for k in range(steps):
            out = plan.evaluate(*draws[(k + 1) % len(draws)])
        dt = time.perf_counter() - t0
        shard_ms = plan.run_timed(steps)
        kernel = plan.summarize_kernel()
        shards = plan.shards
    finally:
        plan.close()
    return {
        "what": "batch.ShardedBatchedGP / clr_sharded_*: %d shards on %d visible GPU(s), devices %s" % (nshards, ndev, devices),
        "batch": B, "shards": [{"device": d, "problems": hi - lo} for d, lo, hi in shards],
        "summarize_kernel_all_shards": kernel,
        "ms_per_step": dt / steps * 1e3, "value": B * steps / dt, "unit": "log-likelihoods/s",
        "device_only_ms_per_step_per_shard": [float(m) / steps for m in shard_ms],
        "set_series_seconds": set_series_s,
        "set_series_note": "host scan of t (max |t|, largest step, warm-up spans) + pageable host->HBM copies of "
                           "t, diag, y on every shard's own thread, concurrently",
        "status_not_ok": int((out[3] != 0).sum()),
    }


def gradient_block(B, N, JR, JC, seed):
    """SURVEY.md 8 row f3 at the headline shape: value + gradient (1 + 2 J_real + 4 J_comp partials) of every problem,
    parallel in n on the resident plan (clr_batch_grad: tangents per (chunk, direction group) from the scanned start
    states + a walk over the chunks), against the sequential tangent kernel (one wave per (problem, partial)) on a
    slice, and one series through the object API both ways."""
    import celerite_amd
    from celerite_amd import batch

    coeffs, t, diag, y = make_inputs(B, N, JR, JC, seed)
    NG = 1 + 2 * JR + 4 * JC
    plan = batch.BatchedGP(B, N, JR, JC)
    try:
        plan.set_series(t, diag, y)
        plan.set_coefficients(*coeffs)
        calls = 3
        timing = {}
        for mode in ("forward", "reverse"):      # (reverse last: it is the default and the one reported as `value`)
            plan.set_grad_mode(mode)
            v, g, st = plan.grad_log_likelihood()
            batch.device_synchronize()
            t0 = time.perf_counter()
            for _ in range(calls):
                v, g, st = plan.grad_log_likelihood()
            timing[mode] = (time.perf_counter() - t0) / calls
            if mode == "forward":
                g_forward = g.copy()
        dt = timing["reverse"]
        info = plan.grad_info()
        fallbacks, chunks = plan.grad_fallbacks(), plan.chunks
        # the accuracy family (sparse: every state stored) through the same plan shape
        coeffs_a, t_a, diag_a, y_a = make_inputs_accuracy(B, N, JR, JC, 4242)
        plan.set_series(t_a, diag_a, y_a)
        plan.set_coefficients(*coeffs_a)
        plan.grad_log_likelihood()
        batch.device_synchronize()
        t0 = time.perf_counter()
        for _ in range(calls):
            va, ga, sta = plan.grad_log_likelihood()
        dt_acc = (time.perf_counter() - t0) / calls
        info_acc = plan.grad_info()
    finally:
        plan.close()

    def sequential(fn):
        batch.set_option("CLR_GRAD_SEQUENTIAL", "1")
        try:
            fn()
            t0 = time.perf_counter()
            out = fn()
            return out, time.perf_counter() - t0
        finally:
            batch.set_option("CLR_GRAD_SEQUENTIAL", None)

    S = min(32, B)
    (vs, gs, sts), dts = sequential(lambda: batch.batch_grad_log_likelihood(*[c[:S] for c in coeffs], t[:S], diag[:S], y[:S]))
    scale = np.maximum(np.abs(gs), 1e-6 * np.max(np.abs(gs), axis=1, keepdims=True))
    e, e2 = np.empty(0), np.empty((0, 0))
    args = (0.0,) + tuple(c[0] for c in coeffs) + (e, e2, e2, t[0], y[0], diag[0])
    sol = celerite_amd.CholeskySolver()
    sol.grad_log_likelihood(*args)
    t0 = time.perf_counter()
    for _ in range(5):
        vo, go = sol.grad_log_likelihood(*args)
    d_obj = (time.perf_counter() - t0) / 5
    (vq, gq), d_seq = sequential(lambda: celerite_amd.CholeskySolver().grad_log_likelihood(*args))
    # widths 16 and 32 through the object API (one series of N samples): the wide scan + chunk-wise forward-mode
    # tangents (csrc/wide_grad_kernels.hip) against the sequential tangent kernel
    wide = {}
    for name, jc in (("width16", 8), ("width32", 16), ("width64", 32)):   # (width 64: round 6, wide_grad_riders64_kernel)
        wc, wt, wd, wy = make_inputs(1, N, 0, jc, seed + 5, d_spread=(jc >= 16))
        wargs = (0.01,) + tuple(c[0] for c in wc) + (e, e2, e2, wt[0], wy[0], wd[0])
        ws = celerite_amd.CholeskySolver()
        ws.grad_log_likelihood(*wargs)
        t0 = time.perf_counter()
        for _ in range(3):
            vw, gw = ws.grad_log_likelihood(*wargs)
        d_w = (time.perf_counter() - t0) / 3
        (vws, gws), d_ws = sequential(lambda: celerite_amd.CholeskySolver().grad_log_likelihood(*wargs))
        wide[name] = {"N": N, "partials": 1 + 4 * jc, "ms_per_call": d_w * 1e3, "sequential_ms_per_call": d_ws * 1e3,
                      "value_rel": abs(vw - vws) / abs(vws), "grad_rel_max": float(np.max(np.abs(gw - gws)) / np.max(np.abs(gws)))}
    return {
        "object_api_wide": wide,
        "workload": "grad_log_likelihood (solver.cpp:347-463): batch=%d x N=%d, width %d, %d partials per problem" % (B, N, JR + 2 * JC, NG),
        "path": "clr_batch_grad, reverse mode: evaluation by the scan, riders + per-sample record per chunk, adjoint walk "
                "over the chunks, one reverse sweep per chunk for all partials",
        "scan_chunks": chunks, "ms_per_call": dt * 1e3, "value": B / dt, "unit": "gradients/s",
        "partials_per_s": B * NG / dt, "sequential_fallbacks": fallbacks, "status_not_ok": int((st != 0).sum()),
        "reverse_sweep": info,
        "forward_mode": {"what": "one tangent per partial from the scanned start states (two per wave) + walk over the chunks",
                         "ms_per_call": timing["forward"] * 1e3,
                         "reverse_vs_forward_rel_max": float(np.max(np.abs(g - g_forward) / np.max(np.abs(g_forward), axis=1, keepdims=True)))},
        "accuracy_family": {"ms_per_call": dt_acc * 1e3, "reverse_sweep": info_acc, "status_not_ok": int((sta != 0).sum()),
                            "note": "sparse series: every step asks for a stored state; every 4th is stored, the others are rebuilt forwards by the sweep (GradStore::span; 21 doubles per sample instead of 54)"},
        "sequential_kernel_slice": {"problems": S, "ms_per_problem_incl_upload": dts * 1e3 / S,
                                    "value_rel_max": rel_err(v[:S], vs),
                                    "grad_rel_max": float(np.max(np.abs(g[:S] - gs) / np.max(np.abs(gs), axis=1, keepdims=True))),
                                    "grad_rel_max_per_partial": float(np.max(np.abs(g[:S] - gs) / scale)),
                                    "note": "one wave per (problem, partial), sequential in n (csrc/grad_kernels.hip).  grad_rel_max: "
                                            "against the problem's largest partial (the tests' measure); ..._per_partial: against each "
                                            "partial's own size (floor 1e-6 of the largest) -- partials near a zero crossing inflate it "
                                            "(round 4's 2.7e-10).  Neither side is off: on the oracle fixture of the headline length "
                                            "(tests/golden/grad_n1e5_w8.json) the plan is 3.9e-12 and this kernel 3.0e-12 from "
                                            "oracle/grad.py per partial (test_plan_gradient_full_size_against_the_oracle_fixture)"},
        "object_api_one_series": {"N": N, "ms_per_call": d_obj * 1e3, "sequential_ms_per_call": d_seq * 1e3,
                                  "grad_rel_max": float(np.max(np.abs(go - gq)) / np.max(np.abs(gq)))},
    }


def object_api_config():
    """BASELINE configs[0]: one series, N = 1000, 1 real + 1 SHO term (width 3) through the
    drop-in object API (GP.compute + GP.log_likelihood), oracle timed the same way."""
    import celerite_amd
    from celerite_amd import terms
    from oracle import ref

    rng = np.random.RandomState(42)
    N = 1000
    t = np.sort(rng.uniform(0, 10, N))
    yerr = rng.uniform(0.1, 0.2, N)
    y = np.sin(t) + yerr * rng.randn(N)
    kernel = terms.RealTerm(log_a=0.1, log_c=0.5) + terms.SHOTerm(log_S0=0.1, log_Q=1.0, log_omega0=0.5)
    gp = celerite_amd.GP(kernel)
    gp.compute(t, yerr)
    ll = gp.log_likelihood(y)
    a_r, c_r, a_c, b_c, c_c, d_c = kernel.coefficients
    empty, empty2 = np.empty(0), np.empty((0, 0))

    def gpu_call():
        gp.solver.compute(0.0, a_r, c_r, a_c, b_c, c_c, d_c, empty, empty2, empty2, t, yerr ** 2)
        return gp.solver.dot_solve(y) + gp.solver.log_determinant()

    r = ref.RefSolver()

    def cpu_call():
        r.compute(0.0, a_r, c_r, a_c, b_c, c_c, d_c, empty, empty2, empty2, t, yerr ** 2)
        return r.dot_solve(y) + r.log_determinant()

    def gpu_hinted():   # as GP.log_likelihood does it: the residual is announced before the factorisation
        gp.solver._hint_rhs(y)
        gp.solver.compute(0.0, a_r, c_r, a_c, b_c, c_c, d_c, empty, empty2, empty2, t, yerr ** 2)
        return gp.solver.dot_solve(y) + gp.solver.log_determinant()

    g, c, gh = gpu_call(), cpu_call(), gpu_hinted()
    tg, tc, th = best_of_3(gpu_call), best_of_3(cpu_call), best_of_3(gpu_hinted)
    ll0 = -0.5 * (c + N * np.log(2 * np.pi))
    W = 3
    return {
        "workload": "BASELINE configs[0]: single series N=1000, RealTerm + SHOTerm (width 3), "
                    "CholeskySolver.compute + dot_solve + log_determinant through the object API",
        "N": N, "width": W, "ms_per_loglik": th * 1e3, "value": 1.0 / th, "unit": "log-likelihoods/s",
        "what": "hint + compute + dot_solve + log_determinant, the calls GP.log_likelihood makes after a parameter "
                "change (the quadratic form rides on the factorisation pass)",
        "separate_compute_and_dot_solve_ms": tg * 1e3,
        "cpu_oracle": {"ms_per_loglik": tc * 1e3, "value": 1.0 / tc, "cores": 1, "timing": "best of 3 (timer.py)"},
        "roofline": {"bound": "launch latency", "note": "one problem of 1.3e5 flop: the call is launch- and "
                     "copy-bound; algorithmic rate below", "achieved": algorithmic_flops_per_loglik(N, W) / th / 1e12,
                     "peak": PEAK_FP64_TFLOPS, "unit": "TFLOP/s",
                     "frac": algorithmic_flops_per_loglik(N, W) / th / 1e12 / PEAK_FP64_TFLOPS},
        "parity": {"loglike_rel": abs(ll - ll0) / abs(ll0), "value_rel": abs(g - c) / abs(c),
                   "hinted_value_rel": abs(gh - c) / abs(c), "tolerance": 1e-10},
    }


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=1024, help="problems per GPU")
    ap.add_argument("--nsamples", type=int, default=100000)
    ap.add_argument("--jreal", type=int, default=2)
    ap.add_argument("--jcomp", type=int, default=3)
    ap.add_argument("--chunks", type=int, default=0, help="scan chunks per problem (0 = auto)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the BASELINE configs 0/1/4 block")
    ap.add_argument("--no-shared-series", action="store_true", help="skip the layout-(ii) leg (profiling runs: keeps the per-kernel counter averages to the distinct-series launches)")
    ap.add_argument("--no-accuracy-family", action="store_true", help="skip the accuracy-family leg")
    ap.add_argument("--no-gradient", action="store_true", help="skip the grad_log_likelihood leg")
    ap.add_argument("--no-object-api", action="store_true", help="skip the single-series CholeskySolver leg (the reference's own benchmark kernels)")
    ap.add_argument("--sharded", type=int, default=2, help="shards of the product's own sharded plan (0: skip the leg)")
    ap.add_argument("--no-config3", action="store_true", help="skip the 8-shard B = 8192 leg (20 GB of host arrays)")
    ap.add_argument("--steady-seconds", type=float, default=2.5)
    ap.add_argument("--settle-seconds", type=float, default=0.5, help="untimed extra warm-up before the K timed steps")
    args = ap.parse_args(argv)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        from celerite_amd import batch as _b

        if _b.device_count() < args.gpus and not os.environ.get("CLR_BENCH_SHARE_GPU"):
            raise SystemExit("--gpus %d but only %d MI355X visible" % (args.gpus, _b.device_count()))
        raise SystemExit(spawn_ranks(args.gpus, sys.argv[1:] if argv is None else list(argv)))

    # the one JSON line must be the only thing on stdout: gloo's C++ prints its "[Gloo] Rank ..." notices there,
    # so with several ranks everything else written to fd 1 goes to stderr and rank 0 prints to the saved fd
    json_fd = None
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        sys.stdout.flush()
        json_fd = os.dup(1)
        os.dup2(2, 1)
    dist = Dist()
    if dist.world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, dist.world))

    from celerite_amd import batch  # raises if the HIP extension is missing

    ndev = batch.device_count()
    if ndev < 1:
        raise SystemExit("bench.py needs an MI355X: libcelerite_hip has no CPU path")
    if dist.local_rank >= ndev and not os.environ.get("CLR_BENCH_SHARE_GPU"):
        raise SystemExit("rank %d has no GPU of its own (%d visible)" % (dist.local_rank, ndev))
    # (CLR_BENCH_SHARE_GPU=1: ranks share the visible GPUs -- only to exercise the multi-rank code path
    #  on a one-GPU box; the numbers of such a run mean nothing)
    B, N, JR, JC = args.batch, args.nsamples, args.jreal, args.jcomp
    W = JR + 2 * JC
    K, Wm = max(args.steps, 1), max(args.warmup, 0)

    try:  # the fp64 rate / clock of this box (rank 0's device stands for all; a few ms, outside every timed region)
        tf, mhz, cyc = batch.measure_fp64(2, 20000, device=dist.local_rank % ndev)   # (every rank on its own GPU)
        FP64_LIVE.update({"tflops": tf, "clock_mhz": mhz, "cycles_per_fma": cyc})
    except Exception:
        pass
    coeffs, t, diag, y = make_inputs(B, N, JR, JC, seed=42 + dist.rank)
    draws = [coeffs] + fresh_draws(coeffs, 7, seed=1042 + dist.rank)
    plan = batch.BatchedGP(B, N, JR, JC, device=dist.local_rank % ndev)
    if args.chunks:
        plan.set_chunks(args.chunks)
    plan.set_series(t, diag, y)          # host -> HBM, outside the timed region
    set_series_first_ms = plan.selection_bounds()["set_series_host_ms"]   # (device buffers + pinned staging allocated)
    plan.set_series(t, diag, y)          # ... and again: what every later batch of series costs
    real_loop(plan, draws, Wm)
    # untimed settling beyond the W warm-up steps: a few steps are not enough for the clocks of an idle
    # device (and the host's pinned-copy path) to reach their steady state; reported as settle_steps
    settle, t_s = 0, time.perf_counter()
    while time.perf_counter() - t_s < args.settle_seconds:
        real_loop(plan, draws, 5, offset=settle)
        settle += 5
    plan.synchronize()
    # HIP events around the DOMINANT kernel only inside the timed region (two event records per step: an event
    # record costs ~5 us of stream time, seven of them were 1.5 % of a step); the other kernels are timed by the
    # same loop right after it, outside the timed region
    plan.set_profiling(2)

    # ---- the timed region: exactly K steps between barrier + device sync -------
    dist.barrier()
    batch.device_synchronize()
    t0 = time.perf_counter()
    real_loop(plan, draws, K, offset=1)
    batch.device_synchronize()
    dist.barrier()
    dt = dist.max(time.perf_counter() - t0)
    dominant_ms, nrec_dom = plan.profile()
    plan.set_profiling(True)
    real_loop(plan, draws, K, offset=1)
    kernel_ms, nrec = plan.profile()
    plan.set_profiling(False)
    kernel_name = plan.summarize_kernel()

    # several ranks: EVERY rank holds the first problems of its own slice against the CPU oracle (the single-rank run
    # does the full parity leg below); the worst deviation over the ranks goes into the line
    rank_parity = None
    if dist.world > 1:
        from oracle import ref as _ref
        nchk = min(2, B)
        plan.set_coefficients(*coeffs)
        _, ld_r, q_r, st_r = plan.log_likelihood()
        _, ld0, q0, st0 = _ref.batch_log_likelihood(0.0, *[c[:nchk] for c in coeffs[:6]], t[:nchk], diag[:nchk], y[:nchk])
        bad = 0.0 if np.array_equal(st_r[:nchk], st0) else 1.0
        rank_parity = {"problems_checked_per_rank": nchk,
                       "logdet_rel_max": dist.max(rel_err(ld_r[:nchk], ld0)), "quad_rel_max": dist.max(rel_err(q_r[:nchk], q0)),
                       "status_equal": dist.max(bad) == 0.0, "tolerance": 1e-10}

    out = None
    if dist.rank == 0:
        per = {k: v / max(nrec, 1) for k, v in kernel_ms.items()}
        # the dominant kernel's time is the one measured INSIDE the timed region
        per_all_events = dict(per)
        per["summarize"] = dominant_ms["summarize"] / max(nrec_dom, 1)
        # parity inputs: the base draw
        plan.set_coefficients(*coeffs)
        ll, ld, q, st = plan.log_likelihood()
        replayed = plan.exact_count()
        # the same kernels back to back without the per-step transfers
        dev_ms, dev_k = plan.run_timed(K, relayout_each_step=False)
        # ... and when every evaluation brings NEW series (the chunk-interleaved copy the role-split
        # summarize reads is rebuilt inside every step; a no-op for the staged layout)
        new_ms, new_k = plan.run_timed(max(K // 2, 1), relayout_each_step=True)
        # a longer steady-state leg of the real loop (lets the driver's utilisation sampler see the device)
        n_steady, t_s = 0, time.perf_counter()
        while time.perf_counter() - t_s < args.steady_seconds:
            real_loop(plan, draws, 10, offset=n_steady)
            n_steady += 10
        steady_dt = time.perf_counter() - t_s
        # A/B legs, for information
        plan.set_exact(True)
        plan.enqueue(); plan.synchronize()
        exact_ms, _ = plan.run_timed(max(K // 2, 1))
        plan.set_exact(False)
        plan.set_coefficients(*coeffs)
        plan.set_summarize_mode(2)      # role-split summarize, two waves per SIMD, lazy decay (clr_split_kernels.h)
        plan.enqueue(); plan.synchronize()
        split_ms, split_k = plan.run_timed(max(K // 2, 1), relayout_each_step=False)
        ll2, ld2, q2, st2 = plan.results()
        plan.set_summarize_mode(1)      # ... without the lazy decay
        plan.enqueue(); plan.synchronize()
        split1_ms, split1_k = plan.run_timed(max(K // 2, 1), relayout_each_step=False)
        plan.set_summarize_mode(0)      # the single-wave kernel (round 1's), staged layout
        plan.enqueue(); plan.synchronize()
        single_ms, single_k = plan.run_timed(max(K // 2, 1), relayout_each_step=False)
        ll3, ld3, q3, st3 = plan.results()
        plan.set_summarize_mode(-1)
        mat_steps = max(K // 4, 2)
        plan.enqueue(materialize=True); plan.synchronize()
        # (series resident and laid out once, as in the timed loop: the optimiser case)
        mat_ms, mat_k = plan.run_timed(mat_steps, materialize=True, relayout_each_step=False)
        # (the 20 GB of factor stores make this leg the one that varies most from box to box and run to run -- 3.8 to
        #  4.9 ms of replay on the round's boxes: five timed runs, the MEDIAN one (by step time) is reported, all are listed)
        mat_all = [(mat_ms, mat_k)]
        for _ in range(4):
            mat_all.append(plan.run_timed(mat_steps, materialize=True, relayout_each_step=False))
        mat_runs = [m / mat_steps for m, _ in mat_all]
        mat_replay_runs = [k["replay"] / mat_steps for _, k in mat_all]
        mat_ms, mat_k = sorted(mat_all, key=lambda mk: mk[0])[len(mat_all) // 2]
        # the batched solve (CholeskySolver::solve for all B problems, parallel in n) on the factor just written: K^-1 y
        solve = {}
        try:
            plan.solve()
            solve["reference_layout_first_solve_device_ms"] = plan.solve_device_ms()   # (forms the chunk maps of this factor)
            x_ref = plan.solve()
            solve["reference_layout_device_ms"] = plan.solve_device_ms()
            # ... and the batched dot_L (CholeskySolver::dot_L, cholesky.h:409-431: what GP.sample draws) of the same factor
            plan.dot_L(y)
            solve["dot_L_reference_layout_device_ms"] = plan.solve_device_ms()
        except Exception as e:
            solve["error"] = repr(e)
        # the LEAN layout (SURVEY.md 8d row A-lean): W and D stored, phi and u regenerated by the consumers
        lean = None
        try:
            plan.set_factor_layout("lean")
            plan.enqueue(materialize=True); plan.synchronize()
            lean_all = [plan.run_timed(mat_steps, materialize=True, relayout_each_step=False) for _ in range(5)]
            lean_ms, lean_k = sorted(lean_all, key=lambda mk: mk[0])[len(lean_all) // 2]
            lld, lq = plan.results()[1:3]
            if "error" not in solve:
                plan.solve()
                solve["lean_layout_first_solve_device_ms"] = plan.solve_device_ms()
                x_lean = plan.solve()
                solve["lean_layout_device_ms"] = plan.solve_device_ms()
                solve["lean_vs_reference_layout_rel"] = float(np.max(np.abs(x_lean - x_ref)) / np.max(np.abs(x_ref)))
                from oracle import ref as _ref2
                _r = _ref2.RefSolver()
                _e, _e2 = np.empty(0), np.empty((0, 0))
                t0_ = time.perf_counter()
                _r.compute(0.0, *[c[0] for c in coeffs], _e, _e2, _e2, t[0], diag[0])
                t1_ = time.perf_counter()
                x0 = _r.solve(y[0])[:, 0]
                solve["cpu_oracle_solve_ms_per_problem"] = (time.perf_counter() - t1_) * 1e3
                solve["problem0_vs_oracle_rel"] = float(np.max(np.abs(x_lean[0] - x0)) / np.max(np.abs(x0)))
                del x_lean, x_ref
                Lz = plan.dot_L(y)
                solve["dot_L_lean_layout_device_ms"] = plan.solve_device_ms()
                Lz0 = _r.dot_L(y[0])[:, 0]
                solve["dot_L_problem0_vs_oracle_rel"] = float(np.max(np.abs(Lz[0] - Lz0)) / np.max(np.abs(Lz0)))
                del Lz
            lean = {"ms": lean_ms / mat_steps, "k": {k: v / mat_steps for k, v in lean_k.items()},
                    "runs": [m / mat_steps for m, _ in lean_all], "bytes_per_problem": plan.factor_bytes(),
                    "logdet_vs_fused_rel": rel_err(lld[st == 0], ld[st == 0]), "quad_vs_fused_rel": rel_err(lq[st == 0], q[st == 0])}
        except Exception as e:  # a failing side leg must not lose the headline line
            lean = {"error": repr(e)}
        plan.set_factor_layout("reference")
        # a survey over many light curves: every step brings NEW series from (pageable) host memory -- wall clock around
        # clr_batch_set_series (staged upload, device scans of t, relayout) + the evaluation + the results
        t0 = time.perf_counter()
        for _ in range(3):
            plan.set_series(t, diag, y)
            plan.enqueue()
            plan.results()
        survey_ms = (time.perf_counter() - t0) / 3 * 1e3

        value = dist.world * B * K / dt
        out = {
            "metric": "GP log-likelihoods/sec, N=1e5 J=8 batch=1024; log_det rel-err vs CPU ref",
            "value": value, "unit": "log-likelihoods/s", "n_gpus": dist.world, "steps": K, "warmup": Wm, "settle_steps": settle,
            "ms_per_step": dt / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {
                "workload": "BASELINE configs[2]: batch=%d problems/GPU x N=%d samples, width J=%d (%d real + %d "
                            "complex terms), fp64, fused log-likelihood (compute + dot_solve + log_determinant), "
                            "chunked scan over N; per step: fresh hyper-parameter draw host->HBM, 5 kernels, "
                            "results HBM->host" % (B, N, W, JR, JC),
                "batch_per_gpu": B, "N": N, "width": W, "J_real": JR, "J_comp": JC,
                "scan_chunks": plan.chunks[0], "chunk_len": plan.chunks[1],
                "parallelism": "batch-sharded x%d, no collective" % dist.world,
                "series_layout": "resident in HBM: the API's row-major arrays plus, for the role-split summarize "
                                 "kernel, a chunk-interleaved copy made once per set_series (outside the timed "
                                 "region: the series do not change between optimiser steps)",
                "summarize_kernel": kernel_name,
                "problems_replayed": replayed,
            },
            "kernels_ms": per, "kernel_events_recorded_steps": nrec,
            "kernels_ms_note": "summarize: HIP events inside the timed region (two event records per step); the other "
                               "kernels: the same real loop repeated right after it with every kernel bracketed "
                               "(seven event records per step, ~5 us each; that loop's summarize time: %.4f ms)"
                               % per_all_events["summarize"],
            "device_only": {"what": "the same kernels back to back, coefficients resident (round 1's `value`)",
                            "ms_per_step": dev_ms / K, "value": B / (dev_ms / K * 1e-3) * dist.world,
                            "kernels_ms": {k: v / K for k, v in dev_k.items()}},
            "new_series_every_step": {"what": "device-only step including the relayout pass of series that are ALREADY in "
                                              "HBM (relayout only: the host scan of t and the host->HBM copies of "
                                              "clr_batch_set_series are not in it; their time for this batch is "
                                              "set_series_host_ms, pageable host memory)",
                                      "ms_per_step": new_ms / max(K // 2, 1),
                                      "value": B / (new_ms / max(K // 2, 1) * 1e-3) * dist.world,
                                      "relayout_ms": new_k["relayout"] / max(K // 2, 1),
                                      "set_series_host_ms": plan.selection_bounds()["set_series_host_ms"],
                                      "set_series_first_call_ms": set_series_first_ms,
                                      "set_series_note": "clr_batch_set_series of 2.46 GB of pageable NumPy arrays: 8 host "
                                                         "threads staging through pinned buffers + device-side scans of t "
                                                         "(csrc/series_io.hip); the first call also allocates",
                                      "step_including_set_series_ms": survey_ms,
                                      "value_including_set_series": B / (survey_ms * 1e-3) * dist.world,
                                      "step_including_set_series_note": "measured wall clock per step of set_series (2.46 GB "
                                                                        "host -> HBM) + evaluation + results, 3 steps"},
            "steady_state": {"seconds": steady_dt, "steps": n_steady, "value": B * n_steady / steady_dt * dist.world},
            "value_steady": B * n_steady / steady_dt * dist.world,
            "timed_region_s": dt,
            "status_not_ok": int((st != 0).sum()),
            "roofline": roofline_block(per, B, N, W, pmc_traffic(max(per, key=per.get), B, N, JR, JC, plan.chunks[0])),
            "ab": {
                "exact_replay_ms_per_step": exact_ms / max(K // 2, 1),
                "summarize_kernels": {
                    "what": "device-only step with each summarize kernel: single wave (one wave per SIMD, staged "
                            "series), role split (two waves per SIMD, clr_split_kernels.h), role split with the "
                            "decay factored out of the state (the default at this shape)",
                    "single_wave": {"ms_per_step": single_ms / max(K // 2, 1),
                                    "summarize_ms": single_k["summarize"] / max(K // 2, 1)},
                    "role_split": {"ms_per_step": split1_ms / max(K // 2, 1),
                                   "summarize_ms": split1_k["summarize"] / max(K // 2, 1)},
                    "role_split_lazy": {"ms_per_step": split_ms / max(K // 2, 1),
                                        "summarize_ms": split_k["summarize"] / max(K // 2, 1)},
                    "lazy_vs_single_logdet_rel": rel_err(ld2[st == 0], ld3[st == 0]),
                    "lazy_vs_single_quad_rel": rel_err(q2[st == 0], q3[st == 0])},
            },
        }
        factor_bytes = B * 8.0 * N * (3 * W + 1)           # phi, u, W, D written
        bytes_ = B * algorithmic_bytes_per_loglik(N)
        mat_replay_s = mat_k["replay"] / mat_steps * 1e-3
        mat_step_s = mat_ms / mat_steps * 1e-3
        out["materialize"] = {
            "what": "same step, additionally writing the factor (phi, u, W, D) of all problems to HBM",
            "ms_per_step": mat_ms / mat_steps, "value": B / mat_step_s * dist.world,
            "kernels_ms": {k: v / mat_steps for k, v in mat_k.items()},
            "ms_per_step_of_each_timed_run": mat_runs, "replay_ms_of_each_timed_run": mat_replay_runs,
            "reported_run": "the median of the five timed runs (by step time)",
            "roofline": {"kernel": "replay (materialising)", "bound": "hbm",
                         "achieved": (factor_bytes + bytes_) / mat_replay_s / 1e9, "peak": PEAK_HBM_GBS,
                         "unit": "GB/s", "frac": (factor_bytes + bytes_) / mat_replay_s / 1e9 / PEAK_HBM_GBS,
                         "bytes_per_launch": factor_bytes + bytes_,
                         "traffic": pmc_traffic("replay (materialising)", B, N, JR, JC, plan.chunks[0]),
                         "whole_step_frac": (factor_bytes + 2 * bytes_) / mat_step_s / 1e9 / PEAK_HBM_GBS,
                         "note": "factor written (8 N (3W+1) B per problem) + t, diag, y read, over the replay "
                                 "kernel's HIP-event time; whole_step_frac = (factor + two passes over the "
                                 "series) over the whole materialising step (summarize runs first)"},
        }
        if lean and "ms" in lean:
            lean_factor = B * float(lean["bytes_per_problem"]) * N / (plan.chunks[0] * plan.chunks[1])  # (without the chunk padding)
            lean_step_s = lean["ms"] * 1e-3
            out["materialize_lean"] = {
                "what": "the same step storing the LEAN factor: W and D only (8 N (W + 1) bytes per problem); phi and u are "
                        "pure functions of (t, coefficients) and are regenerated by the consumers (clr_batch_get_factor "
                        "expands on the device: W, D, u bit-identical to the reference layout's arrays, phi to one ulp)",
                "ms_per_step": lean["ms"], "value": B / lean_step_s * dist.world, "kernels_ms": lean["k"],
                "ms_per_step_of_each_timed_run": lean["runs"], "speedup_vs_reference_layout": (mat_ms / mat_steps) / lean["ms"],
                "factor_GB": lean_factor / 1e9, "factor_GB_reference_layout": factor_bytes / 1e9,
                "logdet_vs_fused_rel": lean["logdet_vs_fused_rel"], "quad_vs_fused_rel": lean["quad_vs_fused_rel"],
                "roofline": {"bound": "fp64_valu", "note": "with 2.8x fewer bytes to store the replay is the plain recurrence (VALU-bound) "
                                                          "and the step two VALU-bound passes; HBM view for information",
                             "whole_step_tflops": 2 * B * algorithmic_flops_per_loglik(N, W) / lean_step_s / 1e12,
                             "whole_step_frac_fp64": 2 * B * algorithmic_flops_per_loglik(N, W) / lean_step_s / 1e12 / PEAK_FP64_TFLOPS,
                             "hbm_whole_step_frac": (lean_factor + 2 * bytes_) / lean_step_s / 1e9 / PEAK_HBM_GBS,
                             "bytes_per_step": lean_factor + 2 * bytes_}}
        elif lean:
            out["materialize_lean"] = lean
        if solve:
            solve["what"] = ("clr_batch_solve: K^-1 y for all %d problems from the materialised factor (CholeskySolver::solve, cholesky.h:218-318, "
                             "as two chunked affine scans; device time of its kernels incl. the two relayouts, right-hand side = the "
                             "plan's resident y); bytes through HBM: four passes over the factor + the relayouts" % B)
            if "lean_layout_device_ms" in solve:
                solve["value_lean"] = B / (solve["lean_layout_device_ms"] * 1e-3)
                solve["unit"] = "solves/s"
            out["batched_solve"] = solve
        # SURVEY.md 8(d) layout (ii): ONE series shared by all B hyper-parameter draws (the MCMC case; t, diag, y with
        # stride 0: 2.4 MB of series in HBM instead of 2.4 GB).  For information; `value` is layout (i), B distinct series.
        try:
            if args.no_shared_series:
                raise RuntimeError("skipped (--no-shared-series)")
            plan.set_series(t[0], diag[0], y[0])
            plan.set_coefficients(*coeffs)
            plan.enqueue(); plan.synchronize()
            sh_ms, sh_k = plan.run_timed(K, relayout_each_step=False)
            lls, lds, qs, sts = plan.results()
            out["shared_series"] = {
                "what": "layout (ii): one series x B draws, device-only step (series stride 0)",
                "ms_per_step": sh_ms / K, "value": B * dist.world / (sh_ms / K * 1e-3),
                "kernels_ms": {k: v / K for k, v in sh_k.items()},
                "problem0_logdet_vs_distinct_series_run": float(abs(lds[0] - ld[0]) / abs(ld[0])),
            }
        except Exception as e:  # a failing side leg must not lose the headline line
            out["shared_series"] = {"error": repr(e)}
        if dist.world == 1 and not args.no_cpu_baseline:
            out.update(cpu_baseline_and_parity(coeffs, t, diag, y, ld, q, st, B, N))
        if rank_parity is not None:
            out["multi_rank_parity"] = rank_parity
    plan.close()
    if dist.rank == 0 and dist.world == 1 and not args.no_accuracy_family:
        try:
            out["accuracy_family"] = accuracy_family_block(B, N, JR, JC, max(K // 2, 5), 8, 4242)
        except Exception as e:  # a failing side leg must not lose the headline line
            out["accuracy_family"] = {"error": repr(e)}
    if dist.rank == 0 and dist.world == 1 and not args.no_object_api:
        try:
            out["object_api"] = object_api_block()
        except Exception as e:  # (a side leg must not cost the headline line)
            out["object_api"] = {"error": repr(e)}
    if dist.rank == 0 and dist.world == 1 and not args.no_gradient:
        try:
            out["gradient"] = gradient_block(B, N, JR, JC, 42)
        except Exception as e:
            out["gradient"] = {"error": repr(e)}
    if dist.rank == 0 and dist.world == 1 and args.sharded > 0:
        try:
            out["sharded_product_path"] = sharded_block(B, N, JR, JC, args.sharded, max(K // 2, 5), 42)
        except Exception as e:
            out["sharded_product_path"] = {"error": repr(e)}
    if dist.rank == 0 and dist.world == 1 and args.sharded > 0 and not args.no_config3:
        # BASELINE configs[3] at its own size through the product's sharded plan: 8 shards x 1024 problems on the
        # visible GPU(s) (on a one-GPU box the eight shards share it: 38 GB of HBM)
        try:
            out["sharded_config3_b8192"] = sharded_block(8 * B, N, JR, JC, 8, 3, 42, tile=8)
        except Exception as e:
            out["sharded_config3_b8192"] = {"error": repr(e)}
    if dist.rank == 0 and dist.world == 1 and not args.no_configs:
        cfg = {}
        for key, fn in [("config0_object_api", object_api_config),
                        ("config1_b256_n1e4_w4", lambda: batch_config(
                            "BASELINE configs[1]: batch=256, N=1e4, width 4 (2 complex terms)", 256, 10000, 0, 2, 20, 64, 7)),
                        ("config4_b256_n1e5_w32", lambda: batch_config(
                            "BASELINE configs[4]: batch=256, N=1e5, width 32 (16 complex terms, log d ~ U(0,3))",
                            256, 100000, 0, 16, 5, 4, 11, d_spread=True)),
                        # not a BASELINE configuration: the reference's dynamic-width arm (cholesky.h:203) at the widest
                        # shape the chunked scan covers since round 5 (widths 33..64 parallel in n; 174 ms sequential)
                        ("extra_b256_n1e5_w64", lambda: batch_config(
                            "extra (not in BASELINE): batch=256, N=1e5, width 64 (32 complex terms, log d ~ U(0,3)): "
                            "chunked scan at the padded width 64", 256, 100000, 0, 32, 3, 2, 288, d_spread=True))]:
            try:
                cfg[key] = fn()
            except Exception as e:  # a failing side leg must not lose the headline line
                cfg[key] = {"error": repr(e)}
        out["configs"] = cfg
    if out is not None:
        promote(out)
    dist.barrier()
    dist.close()
    if out is not None:
        full_path = write_full_record(out)
        line = headline_line(out, full_path)
        sys.stderr.flush()
        if json_fd is None:
            sys.stdout.write(line + "\n")
            sys.stdout.flush()
        else:
            sys.stdout.flush()
            os.write(json_fd, (line + "\n").encode())
    return out


HEADLINE_MAX_BYTES = 6000     # the driver keeps the last 8 KB of stdout: the line it parses must fit with room to spare

# the keys of the driver's contract, in the order they are printed
HEADLINE_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                 "vs_baseline", "dtype", "data", "timed_region_s")
# what may be dropped from `roofline` (last first) when the line would not fit: the promoted side numbers, never the
# dominant kernel's own figures
ROOFLINE_CORE = ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "launch_ms", "algorithmic_flops_per_launch",
                 "algorithmic_bytes_per_launch", "hbm_achieved_GBps", "hbm_frac")
CONFIG_CORE = ("workload", "batch_per_gpu", "N", "width", "J_real", "J_comp", "scan_chunks", "chunk_len", "parallelism",
               "summarize_kernel", "problems_replayed", "value_steady")


def _short(obj, limit=160):
    """Strings cut to `limit` characters, floats to 6 significant digits, recursively (the headline carries numbers;
    the prose lives in the full record)."""
    if isinstance(obj, str):
        return obj if len(obj) <= limit else obj[:limit - 1] + "~"
    if isinstance(obj, bool) or obj is None or isinstance(obj, int):
        return obj
    if isinstance(obj, float):
        if obj != obj or obj in (float("inf"), float("-inf")):
            return None               # strict JSON has no NaN / Infinity
        return float("%.6g" % obj)
    if isinstance(obj, dict):
        return {str(k): _short(v, limit) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return [_short(v, limit) for v in obj]
    try:
        return _short(float(obj), limit)
    except Exception:
        return _short(str(obj), limit)


def headline_record(out, full_path=None):
    """The record of the ONE stdout line: the contract's keys + `config`, `roofline` (with the promoted numbers) and
    `cpu_baseline`.  Everything else of `out` is named in `extra_keys` and kept in the full record."""
    rec = {k: out.get(k) for k in HEADLINE_KEYS}
    rec["config"] = {k: v for k, v in (out.get("config") or {}).items() if k in CONFIG_CORE}
    rec["roofline"] = dict(out.get("roofline") or {})
    cb = out.get("cpu_baseline")
    if cb is not None:
        rec["cpu_baseline"] = cb
    if out.get("multi_rank_parity") is not None:
        rec["multi_rank_parity"] = out["multi_rank_parity"]
    rec["status_not_ok"] = out.get("status_not_ok")
    rec["full_record"] = full_path
    rec["extra_keys"] = sorted(k for k in out if k not in rec)
    return _short(rec)


def headline_line(out, full_path=None, max_bytes=HEADLINE_MAX_BYTES):
    """One strict-JSON line of at most `max_bytes` bytes (tests/test_host_api.py holds it to that)."""
    rec = headline_record(out, full_path)
    line = json.dumps(rec, allow_nan=False, separators=(",", ":"))
    droppable = [k for k in rec["roofline"] if k not in ROOFLINE_CORE]
    while len(line.encode()) > max_bytes and droppable:
        rec["roofline"].pop(droppable.pop())
        line = json.dumps(rec, allow_nan=False, separators=(",", ":"))
    for key in ("extra_keys", "multi_rank_parity"):
        if len(line.encode()) > max_bytes and key in rec:
            rec.pop(key)
            line = json.dumps(rec, allow_nan=False, separators=(",", ":"))
    if len(line.encode()) > max_bytes:
        rec = _short(rec, 60)
        line = json.dumps(rec, allow_nan=False, separators=(",", ":"))
    if len(line.encode()) > max_bytes:
        raise RuntimeError("bench.py: the headline line is %d bytes (> %d)" % (len(line.encode()), max_bytes))
    return line


def write_full_record(out):
    """The complete record (every leg, every note) goes to a side file and to stderr; stdout carries the headline only."""
    text = json.dumps(out)
    path = os.environ.get("CLR_BENCH_FULL")
    if not path:
        d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gpurun_out")
        try:
            os.makedirs(d, exist_ok=True)
            path = os.path.join(d, "bench_full.json")
        except OSError:
            path = os.path.join(tempfile.gettempdir(), "bench_full.json")
    try:
        with open(path, "w") as f:
            f.write(text + "\n")
    except OSError:
        path = None
    sys.stderr.write("bench.py full record:\n" + text + "\n")
    return path


def object_api_block():
    """One series through ``CholeskySolver`` -- the reference's OWN benchmark (examples/benchmark/run.py:37-38, 66-84,
    131-138: real terms (1, 0.1) + identical complex terms (0.1, 2, 1.6), t = sort(rand), yerr ~ U(0.1, 0.2); `compute`
    and `dot_solve` timed apart, best of 3) at N = 65 536 and widths 16 / 128, with the CPU oracle's compute of the
    narrower one beside it (one core; tools/gpu_reference_benchmark_grid.py has the whole grid)."""
    import celerite_amd
    from celerite_amd import terms
    from oracle import ref
    rng = np.random.RandomState(42)
    N = 65536
    t = np.sort(rng.rand(2 ** 19))[:N]
    d = rng.uniform(0.1, 0.2, 2 ** 19)[:N] ** 2
    z = rng.randn(N)
    e_, e2_ = np.empty(0), np.empty((0, 0))

    def best(fn, reps=3):
        b = np.inf
        for _ in range(reps):
            t0 = time.perf_counter(); fn(); b = min(b, time.perf_counter() - t0)
        return b * 1e3

    res = {"N": N, "what": "reference benchmark kernels (examples/benchmark/run.py), one series, ms per call, best of 3"}
    for width in (16, 128):
        j = width // 2
        kernel = terms.RealTerm(1.0, 0.1)
        for _ in range((2 * j - 1) % 2):
            kernel += terms.RealTerm(1.0, 0.1)
        for _ in range((2 * j - 1) // 2):
            kernel += terms.ComplexTerm(0.1, 2.0, 1.6)
        cs = [np.asarray(c, dtype=float) for c in kernel.coefficients]
        s = celerite_amd.CholeskySolver()
        s.compute(0.0, *cs, e_, e2_, e2_, t, d)
        s.dot_solve(z)
        rec = {"compute_ms": best(lambda: s.compute(0.0, *cs, e_, e2_, e2_, t, d))}
        s.dot_solve(z)      # (builds whatever the sweeps keep per factor)
        rec["dot_solve_ms"] = best(lambda: s.dot_solve(z))
        rec["route"] = list(s._route()[:2])
        if width == 16:
            r = ref.RefSolver()
            rec["cpu_compute_ms"] = best(lambda: r.compute(0.0, *cs, e_, e2_, e2_, t, d), 1)
            rec["cpu_dot_solve_ms"] = best(lambda: r.dot_solve(z), 1)
            rec["logdet_rel"] = abs(s.log_determinant() - r.log_determinant()) / abs(r.log_determinant())
            rec["dot_solve_rel"] = abs(s.dot_solve(z) - r.dot_solve(z)) / abs(r.dot_solve(z))
        res["width%d" % width] = rec
    return res


def promote(out):
    """The driver's record keeps `config`, `roofline` and `cpu_baseline` of the line in full and only the NAMES of the
    other keys: the numbers the north star's bars are judged on are copied into `roofline` / `config` (compact, no
    prose)."""
    r, c = out["roofline"], out["config"]
    r["value_steady"] = out.get("value_steady")
    c["value_steady"] = out.get("value_steady")
    ns = out.get("new_series_every_step") or {}
    if "set_series_host_ms" in ns:
        r["set_series_ms"] = {"steady": ns["set_series_host_ms"], "first_call": ns.get("set_series_first_call_ms"),
                              "GBps": 3 * 8e-9 * c["batch_per_gpu"] * c["N"] / (ns["set_series_host_ms"] * 1e-3)}
    m = out.get("materialize")
    if m and "roofline" in m:
        mr = m["roofline"]
        r["materialize"] = {"bound": "hbm", "frac": mr["frac"], "whole_step_frac": mr["whole_step_frac"],
                            "achieved_GBps": mr["achieved"], "peak_GBps": mr["peak"],
                            "replay_ms": m["kernels_ms"]["replay"], "step_ms": m["ms_per_step"],
                            "bytes_per_launch": mr["bytes_per_launch"], "traffic": mr.get("traffic")}
        r["materialize_frac"] = mr["frac"]
        r["materialize_whole_step_frac"] = mr["whole_step_frac"]
    bs = out.get("batched_solve")
    if bs and "lean_layout_device_ms" in bs:
        r["batched_solve_ms"] = {"reference_layout": bs.get("reference_layout_device_ms"), "lean_layout": bs["lean_layout_device_ms"],
                                 "problem0_vs_oracle_rel": bs.get("problem0_vs_oracle_rel")}
        if "dot_L_lean_layout_device_ms" in bs:
            r["batched_dot_L_ms"] = {"reference_layout": bs.get("dot_L_reference_layout_device_ms"), "lean_layout": bs["dot_L_lean_layout_device_ms"],
                                     "problem0_vs_oracle_rel": bs.get("dot_L_problem0_vs_oracle_rel")}
    ml = out.get("materialize_lean")
    if ml and "ms_per_step" in ml:
        r["materialize_lean"] = {"step_ms": ml["ms_per_step"], "replay_ms": ml["kernels_ms"]["replay"],
                                 "speedup_vs_reference_layout": ml["speedup_vs_reference_layout"], "factor_GB": ml["factor_GB"],
                                 "whole_step_frac_fp64": ml["roofline"]["whole_step_frac_fp64"],
                                 "hbm_whole_step_frac": ml["roofline"]["hbm_whole_step_frac"]}
    p = out.get("parity")
    if p:
        r["parity"] = {k: p[k] for k in ("logdet_rel_max", "quad_rel_max", "problems_checked", "status_equal") if k in p}
    cfgs = out.get("configs") or {}
    c4 = cfgs.get("config4_b256_n1e5_w32")
    if c4 and "roofline" in c4:
        r["config4"] = {"ms_per_step": c4.get("ms_per_step"), "kernel": c4["roofline"]["kernel"],
                        "launch_ms": c4["roofline"]["launch_ms"], "frac": c4["roofline"]["frac"],
                        "whole_step_frac": c4["roofline"]["whole_step"]["frac_fp64"],
                        "routes": (c4.get("levels") or {}).get("histogram"),
                        "logdet_rel_max": (c4.get("parity") or {}).get("logdet_rel_max")}
        bs4 = c4.get("batched_solve") or {}
        if "solve_device_ms" in bs4:
            r["config4"]["batched_solve_ms"] = bs4["solve_device_ms"]
            r["config4"]["batched_solve_problem0_vs_oracle_rel"] = bs4["problem0_vs_oracle_rel"]
            r["config4"]["materialising_step_ms"] = bs4["materialising_step_ms"]
    c1 = cfgs.get("config1_b256_n1e4_w4")
    if c1:
        r["config1_ms_per_step"] = c1.get("ms_per_step")
    c64 = cfgs.get("extra_b256_n1e5_w64")
    if c64 and "ms_per_step" in c64:
        r["extra_w64"] = {"ms_per_step": c64["ms_per_step"], "scan_chunks": c64.get("scan_chunks"),
                          "logdet_rel_max": (c64.get("parity") or {}).get("logdet_rel_max")}
    for key, name in (("sharded_product_path", "sharded_2x512"), ("sharded_config3_b8192", "sharded_8x1024")):
        sblk = out.get(key)
        if sblk and "value" in sblk:
            r[name] = {"value": sblk["value"], "ms_per_step": sblk["ms_per_step"], "batch": sblk["batch"],
                       "status_not_ok": sblk["status_not_ok"]}
    g = out.get("gradient")
    if g and "ms_per_call" in g:
        r["gradient_ms_per_call"] = g["ms_per_call"]
        if "object_api_wide" in g:
            r["gradient_object_api_ms"] = {k: v["ms_per_call"] for k, v in g["object_api_wide"].items()}
    a = out.get("accuracy_family")
    if a and "ms_per_step" in a:
        r["accuracy_family_ms_per_step"] = a["ms_per_step"]
    oa = out.get("object_api")
    if oa and "width16" in oa:
        r["object_api_ms"] = {k: {kk: oa[k][kk] for kk in ("compute_ms", "dot_solve_ms", "cpu_compute_ms", "cpu_dot_solve_ms") if kk in oa[k]}
                              for k in ("width16", "width128") if k in oa}


def cpu_baseline_and_parity(coeffs, t, diag, y, ld_gpu, q_gpu, st_gpu, B, N):
    """Times the CPU oracle (a like-for-like port of the reference's cholesky.h loops; the
    reference itself needs Eigen and cannot be built here) on this box's host cores with the
    reference's own protocol (celerite/timer.py: best of 3) and checks the GPU results against it."""
    from oracle import ref

    S = min(B, 256)
    sub = [c[:S] for c in coeffs]
    times, res = [], None
    for _ in range(3):
        t0 = time.perf_counter()
        res = ref.batch_log_likelihood(0.0, *sub, t[:S], diag[:S], y[:S], nthreads=1)
        times.append(time.perf_counter() - t0)
    l0, d0, q0, s0 = res
    t1 = min(times)
    cores = os.cpu_count() or 1
    S2 = min(B, max(cores * 4, 64))
    sub2 = [c[:S2] for c in coeffs]
    times2 = []
    for _ in range(3):
        t0 = time.perf_counter()
        ref.batch_log_likelihood(0.0, *sub2, t[:S2], diag[:S2], y[:S2], nthreads=cores)
        times2.append(time.perf_counter() - t0)
    t2 = min(times2)
    ok = s0 == 0
    # the same source compiled for THIS machine ("-O3 -march=native"; the shipped library is built with
    # -march=x86-64-v2 so that it runs on whatever host the GPU box has): both are reported, `value` is the faster
    portable = {"value": S / t1, "flags": "gcc -O3 -march=x86-64-v2 -mtune=generic -ffp-contract=off (oracle/Makefile; built in the build container)",
                "seconds_per_pass": t1}
    native = {"error": "not attempted"}
    try:
        nlib, nflags = ref.load_native()
        if nlib is None:
            native = {"error": nflags}
        else:
            tn = []
            for _ in range(3):
                t0 = time.perf_counter()
                rn = ref.batch_log_likelihood(0.0, *sub, t[:S], diag[:S], y[:S], nthreads=1, lib=nlib)
                tn.append(time.perf_counter() - t0)
            native = {"value": S / min(tn), "flags": "gcc " + nflags + " (built on this box by bench.py: oracle/Makefile, target native)",
                      "seconds_per_pass": min(tn),
                      "logdet_rel_vs_portable_build": rel_err(rn[1][ok], d0[ok]), "status_equal": bool(np.array_equal(rn[3], s0))}
    except Exception as e:  # the baseline must not lose the line
        native = {"error": repr(e)}
    best = native if native.get("value", 0.0) > portable["value"] else portable
    return {
        "cpu_baseline": {
            "value": best["value"], "unit": "log-likelihoods/s", "cores": 1, "kind": "port",
            "flags": best["flags"],
            "sample": "the first %d of the %d problems of the GPU batch (N=%d, width 8), oracle/celerite_ref.c, "
                      "1 thread, best of 3 passes (%.1f s each)" % (S, B, N, best["seconds_per_pass"]),
            "portable_build": {k: v for k, v in portable.items() if k != "seconds_per_pass"},
            "native_build": {k: v for k, v in native.items() if k != "seconds_per_pass"},
            "all_cores": {"value": S2 / t2, "cores": cores, "build": "portable",
                          "sample": "%d problems, one per thread over %d threads, best of 3 (%.1f s each)"
                                    % (S2, cores, t2)},
        },
        "parity": {
            "logdet_rel_max": rel_err(ld_gpu[:S][ok], d0[ok]), "quad_rel_max": rel_err(q_gpu[:S][ok], q0[ok]),
            "status_equal": bool(np.array_equal(st_gpu[:S], s0)),
            "problems_checked": int(S), "tolerance": 1e-10,
        },
    }


if __name__ == "__main__":
    main()
