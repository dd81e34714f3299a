# -*- coding: utf-8 -*-
"""`-m gpu`: the multi-GPU paths under the driver's own test run (contiguous slices of the batch axis,
one process -- or one host thread -- per GPU, no collective on the data path).

  * ``bench.py --gpus 2`` exactly as the driver launches it for N > 1 (``python -m torch.distributed.run ...``), here
    with both ranks on the visible GPU(s) (``CLR_BENCH_SHARE_GPU=1`` on a one-GPU box): ONE JSON line on stdout,
    ``n_gpus == 2``, whole-job value = problems of both ranks over the max-over-ranks time, every rank's slice held
    against the CPU oracle;
  * the product's sharded plan over EVERY visible device (``range(device_count())``: one shard on a one-GPU box, the
    real thing on the 8-GPU node): bit-identical to the unsharded plan, oracle parity at 1e-10.
"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from celerite_amd import batch
from oracle import ref
from _cases import synthetic, coeffs_of

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REL = 1e-10


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.timeout(900)
def test_bench_two_ranks_as_the_driver_launches_it():
    ndev = batch.device_count()
    env = dict(os.environ)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    if ndev < 2:
        env["CLR_BENCH_SHARE_GPU"] = "1"     # two ranks on the one GPU of the box: the code path, not the numbers
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"),
           "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "64", "--nsamples", "10000",
           "--steady-seconds", "0.2", "--settle-seconds", "0.1", "--no-configs", "--no-shared-series"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=840)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, r.stdout.decode()[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["scaling"] == "weak"
    assert out["unit"] == "log-likelihoods/s" and out["dtype"] == "f64"
    assert out["config"]["batch_per_gpu"] == 64 and "x2" in out["config"]["parallelism"]
    # whole-job aggregate: the problems of BOTH ranks over the max-over-ranks time of the K steps
    # (the headline line carries six significant digits)
    assert abs(out["value"] - 2 * 64 * 3 / out["timed_region_s"]) <= 1e-4 * out["value"]
    assert abs(out["ms_per_step"] - out["timed_region_s"] / 3 * 1e3) <= 1e-4 * out["ms_per_step"]
    assert out["status_not_ok"] == 0
    par = out["multi_rank_parity"]
    assert par["status_equal"] and par["logdet_rel_max"] <= REL and par["quad_rel_max"] <= REL, par
    assert "roofline" in out and out["roofline"]["frac"] > 0.0
    assert len(lines[0].encode()) < 6000


@pytest.mark.timeout(900)
def test_bench_one_gpu_line_as_the_driver_parses_it():
    """``python bench.py --gpus 1 --steps 20 --warmup 5`` -- the driver's round-end command.  The driver keeps the last
    8 KB of stdout and parses the LAST line: that line must be strict JSON, well under 8 KB, and carry the headline,
    ``config.workload``, ``roofline`` and ``cpu_baseline`` (round 5's 26.5 KB line came back ``parsed: null``).  The
    complete record goes to a side file."""
    env = dict(os.environ)
    full = os.path.join(ROOT, "gpurun_out", "bench_full_test.json")
    env["CLR_BENCH_FULL"] = full
    os.makedirs(os.path.dirname(full), exist_ok=True)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=840)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    stdout = r.stdout.decode()
    assert len(stdout.encode()) < 8000, len(stdout)
    last = stdout.strip().splitlines()[-1]
    assert len(last.encode()) < 6000

    def no_constants(name):
        raise ValueError("not strict JSON: %s" % name)

    out = json.loads(last, parse_constant=no_constants)
    assert out["n_gpus"] == 1 and out["steps"] == 20 and out["warmup"] == 5
    assert out["metric"].startswith("GP log-likelihoods/sec") and out["unit"] == "log-likelihoods/s"
    assert out["higher_is_better"] is True and out["dtype"] == "f64" and out["vs_baseline"] is None
    assert abs(out["value"] - 1024 * 20 / out["timed_region_s"]) <= 1e-4 * out["value"]
    assert "configs[2]" in out["config"]["workload"] and out["config"]["batch_per_gpu"] == 1024
    roof = out["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "launch_ms"):
        assert key in roof, key
    assert 0.0 < roof["frac"] < 1.0 and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-4
    assert roof["launch_ms"] < out["ms_per_step"]
    assert roof["parity"]["status_equal"] and roof["parity"]["logdet_rel_max"] <= REL and roof["parity"]["quad_rel_max"] <= REL
    cpu = out["cpu_baseline"]
    assert cpu["value"] > 0 and cpu["cores"] == 1 and cpu["kind"] == "port" and cpu["sample"]
    assert out["status_not_ok"] == 0
    whole = json.load(open(full))
    assert whole["value"] == pytest.approx(out["value"], rel=1e-5) and "configs" in whole and "materialize" in whole


def test_sharded_plan_over_every_visible_device():
    """``ShardedBatchedGP`` with one shard per visible device (distinct devices, not one device listed several times):
    on the 8-GPU node this is BASELINE configs[3]'s partitioning; on a one-GPU box a single shard.  Bit-identical to
    the unsharded plan at the same chunk count, oracle parity on every problem, and a second evaluation through
    ``evaluate`` (new coefficients in, results out) agrees with the oracle too."""
    ndev = batch.device_count()
    B, N, JR, JC = max(2 * ndev + 1, 8), 6000, 2, 3
    case = synthetic(B, N, JR, JC, "bench", seed=314)
    l0, d0, q0, s0 = ref.batch_log_likelihood(0.0, *coeffs_of(case), case["t"], case["diag"], case["y"])
    plan = batch.BatchedGP(B, N, JR, JC)
    try:
        plan.set_chunks(32)
        plan.set_series(case["t"], case["diag"], case["y"])
        plan.set_coefficients(*coeffs_of(case))
        want = plan.log_likelihood()
    finally:
        plan.close()
    sp = batch.ShardedBatchedGP(B, N, JR, JC, devices=list(range(ndev)))
    try:
        assert [d for d, _, _ in sp.shards] == list(range(ndev))
        assert [(lo, hi) for _, lo, hi in sp.shards] == [batch.shard_bounds(B, ndev, s) for s in range(ndev)]
        sp.set_chunks(32)
        sp.set_series(case["t"], case["diag"], case["y"])
        got = sp.evaluate(*coeffs_of(case))
        for a, b in zip(want, got):
            assert np.array_equal(a, b, equal_nan=True)
        assert np.array_equal(got[3], s0)
        assert np.max(np.abs(got[1] - d0) / np.abs(d0)) <= REL and np.max(np.abs(got[2] - q0) / np.abs(q0)) <= REL
        assert sp.rescued() == 0                      # (a benign batch: nothing on the checked route)
        sp.set_rescue(0)                              # the inline replay on every shard: a setter that reaches all plans
        got0 = sp.evaluate(*coeffs_of(case))
        for a, b in zip(want, got0):
            assert np.array_equal(a, b, equal_nan=True)
        sp.set_rescue(-1)
        other = synthetic(B, N, JR, JC, "bench", seed=315)
        got2 = sp.evaluate(*coeffs_of(other))
        _, d1, q1, s1 = ref.batch_log_likelihood(0.0, *coeffs_of(other), case["t"], case["diag"], case["y"])
        assert np.array_equal(got2[3], s1)
        assert np.max(np.abs(got2[1] - d1) / np.abs(d1)) <= REL and np.max(np.abs(got2[2] - q1) / np.abs(q1)) <= REL
    finally:
        sp.close()
