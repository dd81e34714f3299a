# -*- coding: utf-8 -*-
"""Host-side logic that needs no GPU: the modelling protocol, the GP front-end's
bookkeeping and error contract, the compiled module's import surface, that the
C-ABI library exports every symbol include/celerite_hip.h declares, and that
the product fails LOUDLY (no CPU fallback) when no MI355X is present."""
import ctypes
import os
import pickle

import numpy as np
import pytest

import celerite_amd
from celerite_amd import GP, batch, modeling, solver, terms

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NO_GPU = batch.device_count() == 0
E = np.empty(0)
NOGEN = (np.empty(0), np.empty((0, 0)), np.empty((0, 0)))


def test_public_surface():  # celerite/__init__.py:20-33
    for name in ("terms", "solver", "modeling", "GP", "CholeskySolver", "__library_version__"):
        assert hasattr(celerite_amd, name)
    assert celerite_amd.__library_version__ == "0.3.0"  # version.h:4-13
    for name in ("get_library_version", "has_autodiff", "LinAlgError", "get_kernel_value",
                 "get_psd_value", "check_coefficients", "CholeskySolver", "CARMASolver"):
        assert hasattr(solver, name)
    s = solver.CholeskySolver()
    for meth in ("compute", "solve", "dot_solve", "dot_L", "dot", "predict", "log_determinant",
                 "computed", "grad_log_likelihood", "__getstate__", "__setstate__"):
        assert hasattr(s, meth)
    assert issubclass(solver.LinAlgError, Exception)


def test_abi_exports_every_declared_symbol():
    import __graft_entry__ as entry

    lib = ctypes.CDLL(batch.LIB_PATH)
    declared = entry.declared_symbols()
    assert len(declared) >= 30
    missing = [s for s in declared if not hasattr(lib, s)]
    assert missing == []
    lib.clr_version.restype = ctypes.c_char_p
    assert lib.clr_version() == b"0.3.0"
    lib.clr_status_string.restype = ctypes.c_char_p
    # what() strings of exceptions.h:14-36
    assert lib.clr_status_string(1) == b"dimension mismatch"
    assert lib.clr_status_string(2) == b"failed to factorize or solve matrix"
    assert lib.clr_status_string(3) == b"you must call 'compute' first"


def test_no_product_code_touches_the_oracle():
    """The oracle is test infrastructure: nothing under celerite_amd/ may import,
    link or mention it."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "celerite_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "celerite_ref" not in text and "from oracle" not in text \
                    and "import oracle" not in text, os.path.join(dirpath, f)


def test_solver_state_machine_without_compute():
    s = solver.CholeskySolver()
    assert s.computed() is False
    with pytest.raises(RuntimeError, match="you must call 'compute' first"):
        s.log_determinant()
    # the reference checks the row count before `computed` (cholesky.h:327-328)
    with pytest.raises(RuntimeError):
        s.dot_solve(np.ones(3))
    # dimension checks come before any device work (cholesky.h:59-69)
    with pytest.raises(RuntimeError, match="dimension mismatch"):
        s.compute(0.0, np.ones(1), np.ones(2), E, E, E, E, *NOGEN, np.arange(4.0), np.ones(4))
    with pytest.raises(RuntimeError, match="dimension mismatch"):
        s.compute(0.0, np.ones(1), np.ones(1), E, E, E, E, *NOGEN, np.arange(4.0), np.ones(5))
    assert s.computed() is False
    s2 = pickle.loads(pickle.dumps(s, -1))  # un-computed round trip, test_celerite.py:270-272
    assert s2.computed() is False
    fresh = solver.CholeskySolver.__new__(solver.CholeskySolver)  # what pickle.loads does
    with pytest.raises(RuntimeError, match="Invalid state"):  # solver.cpp:649
        fresh.__setstate__((1, 2, 3))


@pytest.mark.skipif(not NO_GPU, reason="only meaningful on a box without a GPU")
def test_fails_loudly_without_a_gpu():
    s = solver.CholeskySolver()
    with pytest.raises(RuntimeError, match="no gfx950"):
        s.compute(0.0, np.ones(1), np.ones(1), E, E, E, E, *NOGEN, np.arange(4.0), np.ones(4))
    with pytest.raises(RuntimeError, match="no gfx950|clr_batch_create failed"):
        batch.BatchedGP(2, 10, 1, 0)
    gp = GP(terms.RealTerm(0.1, 0.5))
    with pytest.raises(RuntimeError, match="no gfx950"):
        gp.compute(np.arange(5.0), 0.1)


def test_host_helpers():
    tau = np.array([[0.0, 0.5], [1.0, -2.0]])
    a, c = np.array([1.5, 0.1]), np.array([1.0, 0.3])
    ac, bc, cc, dc = np.array([1.0]), np.array([0.1]), np.array([1.0]), np.array([1.0])
    k = solver.get_kernel_value(a, c, ac, bc, cc, dc, tau)
    at = np.abs(tau)
    want = sum(ai * np.exp(-ci * at) for ai, ci in zip(a, c)) + np.exp(-cc[0] * at) * (
        ac[0] * np.cos(dc[0] * at) + bc[0] * np.sin(dc[0] * at))
    assert k.shape == tau.shape and np.allclose(k, want, rtol=1e-15)
    assert solver.check_coefficients(a, c, ac, bc, cc, dc) is True
    assert solver.check_coefficients(a, c[:1], ac, bc, cc, dc) is False  # utils.h:41
    with pytest.raises(RuntimeError, match="dimension mismatch"):     # q >= p, carma.h:59
        solver.CARMASolver(-0.5, np.array([0.1]), np.array([0.2]))
    assert solver.has_autodiff() is True   # forward-mode gradient kernels (csrc/grad_kernels.hip)


def test_build_gp():  # tests/test_celerite.py:292-309
    kernel = terms.RealTerm(0.5, 0.1)
    kernel += terms.ComplexTerm(0.6, 0.7, 1.0)
    gp = GP(kernel)
    assert gp.vector_size == 5
    assert np.allclose(gp.get_parameter_vector(), [0.5, 0.1, 0.6, 0.7, 1.0])
    gp.set_parameter_vector([0.5, 0.8, 0.6, 0.7, 2.0])
    assert np.allclose(gp.get_parameter_vector(), [0.5, 0.8, 0.6, 0.7, 2.0])
    with pytest.raises(ValueError):
        gp.set_parameter_vector([0.5, 0.8, -0.6])
    with pytest.raises(ValueError):
        gp.set_parameter_vector("face1")


def test_gp_bookkeeping_and_errors():
    gp = GP(terms.RealTerm(0.1, 0.5), mean=1.5, fit_mean=True)
    assert gp.get_parameter_names() == ("kernel:log_a", "kernel:log_c", "mean:value")
    assert gp.computed is False and gp.dirty is True
    with pytest.raises(RuntimeError, match="you must call 'compute' first"):
        gp.log_likelihood(np.ones(3))  # tests/test_celerite.py:341-344
    with pytest.raises(ValueError, match="sorted"):
        gp.compute(np.array([0.0, 2.0, 1.0]), 0.1)  # :358-359
    with pytest.raises(ValueError, match="dimension mismatch"):
        gp.compute(np.zeros((3, 2)), 0.1)
    with pytest.raises(RuntimeError):
        gp.get_matrix()
    K = gp.get_matrix(np.array([0.0, 1.0]), np.array([0.0, 0.5, 1.0]))
    assert K.shape == (2, 3) and np.isclose(K[0, 0], np.exp(0.1))
    with pytest.raises(RuntimeError, match="you must call 'compute' first"):
        gp.grad_log_likelihood(np.ones(3))  # celerite.py:256 -> _process_input
    gp["kernel:log_a"] = 0.3
    assert gp.get_parameter("kernel:log_a") == 0.3
    gp.freeze_parameter("kernel:log_c")
    assert gp.vector_size == 2
    with pytest.warns(UserWarning):
        gp2 = GP(terms.RealTerm(0.1, 0.5), log_white_noise=0.2)
    assert np.isclose(gp2.kernel.jitter, np.exp(0.4)) and gp2.vector_size == 2


def test_modeling_protocol():
    class Line(modeling.Model):
        parameter_names = ("m", "b")

        def get_value(self, x):
            return self.m * x + self.b

        def compute_gradient(self, x):
            return np.array([x, np.ones_like(x)])

    m = Line(m=2.0, b=1.0, bounds=dict(m=(0, 5)))
    assert m.full_size == 2 and len(m) == 2 and m.dirty
    m.dirty = False
    m.set_parameter_vector([3.0, 0.5])
    assert m.dirty and np.allclose(m.get_value(np.array([1.0])), 3.5)
    m.freeze_parameter("b")
    assert m.get_parameter_names() == ("m",) and m.get_gradient(np.ones(2)).shape == (1, 2)
    assert m.get_parameter_bounds(include_frozen=True) == [(0, 5), (None, None)]
    m.set_parameter("m", 9.0)
    assert m.log_prior() == -np.inf
    with pytest.raises(ValueError):
        Line(1.0)
    with pytest.raises(ValueError):
        Line(1.0, 2.0, b=3.0)
    with pytest.raises(ValueError):
        Line(m=1.0)
    with pytest.raises(ValueError):
        Line(m=1.0, b=2.0, c=3.0)
    ms = modeling.ModelSet([("one", Line(1.0, 2.0)), ("two", modeling.ConstantModel(3.0))])
    assert ms.get_parameter_names() == ("one:m", "one:b", "two:value")
    assert ms.get_parameter_dict()["two:value"] == 3.0
    ms.freeze_parameter("one:b")
    assert ms.vector_size == 2 and np.array_equal(ms.unfrozen_mask, [True, False, True])
    with pytest.raises(ValueError):
        ms.freeze_parameter("nope:x")


def test_batch_front_end_validation():
    k = terms.RealTerm(0.1, 0.5) + terms.ComplexTerm(0.6, 0.7, 1.0)
    draws = np.array([[0.1, 0.5, 0.6, 0.7, 1.0], [0.2, 0.4, 0.5, 0.6, 0.9]])
    ar, cr, ac, bc, cc, dc, jit = batch.kernel_coefficient_table(k, draws)
    assert ar.shape == (2, 1) and ac.shape == (2, 1) and np.allclose(jit, 0)
    assert np.allclose(ar[:, 0], np.exp(draws[:, 0])) and np.allclose(dc[:, 0], np.exp(draws[:, 4]))
    assert np.allclose(k.get_parameter_vector(), [0.1, 0.5, 0.6, 0.7, 1.0])  # restored
    sho = terms.SHOTerm(log_S0=0.1, log_Q=0.0, log_omega0=0.5)
    with pytest.raises(ValueError):  # Q crosses 1/2: term count changes between draws
        batch.kernel_coefficient_table(sho, np.array([[0.1, 1.0, 0.5], [0.1, -1.0, 0.5]]))


def test_coefficient_jacobian_table_and_chain_rule():
    """batch.kernel_coefficient_jacobian_table / chain_gradient (host arithmetic): shapes, the order of the
    coefficient columns (that of the batched gradient: jitter | a_real c_real a_comp b_comp c_comp d_comp), restored
    parameters, and the chain rule on a linear test functional of the coefficients."""
    k = terms.RealTerm(0.1, 0.5) + terms.ComplexTerm(0.6, 0.7, 1.0) + terms.JitterTerm(log_sigma=-1.0)
    rng = np.random.RandomState(5)
    draws = k.get_parameter_vector()[None, :] + 0.1 * rng.randn(7, 6)
    jac, jit_jac = batch.kernel_coefficient_jacobian_table(k, draws)
    assert jac.shape == (7, 6, 6) and jit_jac.shape == (7, 6)
    assert np.allclose(k.get_parameter_vector(), [0.1, 0.5, 0.6, 0.7, 1.0, -1.0])
    tab = batch.kernel_coefficient_table(k, draws)
    # every coefficient is exp(one parameter): the Jacobian is a permutation of diag(coefficients); b_comp = 0
    for b in range(7):
        coeffs = np.concatenate([tab[i][b] for i in range(6)])
        assert np.allclose(jac[b].sum(axis=0), coeffs) and np.count_nonzero(jac[b]) == 5
        assert np.allclose(jit_jac[b], [0, 0, 0, 0, 0, 2 * tab[6][b]])
    w = rng.randn(7, 7)     # a functional f = w0 jitter + w[1:] . coefficients has d f / d coefficients = w
    g = batch.chain_gradient(w, jac, jit_jac)
    eps = 1e-6
    for p in range(6):
        hi, lo = draws.copy(), draws.copy()
        hi[:, p] += eps
        lo[:, p] -= eps
        f = []
        for d in (hi, lo):
            t = batch.kernel_coefficient_table(k, d)
            f.append(w[:, 0] * t[6] + np.sum(w[:, 1:] * np.concatenate([t[i] for i in range(6)], axis=1), axis=1))
        assert np.allclose(g[:, p], (f[0] - f[1]) / (2 * eps), rtol=1e-6, atol=1e-8)


def test_shard_bounds_and_sharded_plan_without_a_gpu():
    """clr_shard_bounds is host arithmetic; a sharded plan fails loudly without a device."""
    from celerite_amd import batch
    for total, S in [(1024, 8), (8192, 8), (10, 3), (7, 7), (5, 1)]:
        cuts = [batch.shard_bounds(total, S, s) for s in range(S)]
        assert cuts[0][0] == 0 and cuts[-1][1] == total
        assert all(cuts[i][1] == cuts[i + 1][0] for i in range(S - 1))
        sizes = [hi - lo for lo, hi in cuts]
        assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)
    with pytest.raises(ValueError):
        batch.shard_bounds(10, 3, 3)
    if batch.device_count() == 0:
        with pytest.raises(RuntimeError, match="no gfx950"):
            batch.ShardedBatchedGP(8, 100, 1, 1, devices=[0, 0])


def test_carma_model_algebra_runs_on_the_host():
    """CARMASolver's constructor and get_celerite_coeffs are parameter algebra (carma.h:54-165): no device
    needed, same arrays as the oracle's restatement; log_likelihood is a device kernel and fails loudly."""
    from oracle import carma
    for log_sigma, ar, ma in [(-0.5, [0.1, 0.05, 0.01], [0.2, 0.1]), (0.3, [0.5, -0.2], [0.1]),
                              (0.0, [1.0, 0.3, -0.4, 0.2, 0.05], [0.3, -0.1, 0.2]), (-1.0, [0.4], [])]:
        s = solver.CARMASolver(log_sigma, np.array(ar), np.array(ma, dtype=float))
        o = carma.CARMASolver(log_sigma, ar, ma)
        got, want = s.get_celerite_coeffs(), o.get_celerite_coeffs()
        assert len(got) == 6
        for g, w in zip(got, want):
            assert g.shape == w.shape
            assert np.allclose(g, w, rtol=1e-12, atol=1e-300)
    if NO_GPU:
        with pytest.raises(RuntimeError, match="no gfx950"):
            s.log_likelihood(np.zeros(3), np.zeros(3), np.ones(3))


def test_device_memory_needs_a_gpu():
    if NO_GPU:
        with pytest.raises(RuntimeError, match="no gfx950"):
            batch.device_memory()
    else:
        free, total = batch.device_memory()
        assert 0 < free <= total


def test_bench_headline_line_is_small_strict_json():
    """The driver keeps the last 8 KB of bench.py's stdout and parses the last line (round 5's 26.5 KB line came back
    ``parsed: null``).  The headline built from a real full record (round 5's, ``profiles/r05end_bench.json``), from the
    same record with every string blown up and non-finite numbers in it, and from a minimal one must be ONE strict-JSON
    line below 6000 bytes that carries the contract's keys, ``config.workload``, ``roofline`` and ``cpu_baseline``."""
    import json

    import bench

    full = json.load(open(os.path.join(ROOT, "profiles", "r05end_bench.json")))
    assert len(json.dumps(full)) > 20000          # (the canned record IS the one the driver could not parse)

    def no_constants(name):
        raise ValueError("not strict JSON: %s" % name)

    def check(record):
        line = bench.headline_line(record, "gpurun_out/bench_full.json")
        assert "\n" not in line and len(line.encode()) < bench.HEADLINE_MAX_BYTES <= 6000
        out = json.loads(line, parse_constant=no_constants)
        for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                    "vs_baseline", "dtype", "data", "config", "roofline"):
            assert key in out, key
        assert out["value"] == pytest.approx(record["value"], rel=1e-5)
        assert out["config"]["workload"].startswith("BASELINE configs[2]")
        for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
            assert key in out["roofline"], key
        assert out["roofline"]["frac"] == pytest.approx(record["roofline"]["frac"], rel=1e-5)
        return out

    out = check(full)
    assert out["cpu_baseline"]["cores"] == 1 and out["cpu_baseline"]["kind"] == "port" and out["cpu_baseline"]["value"] > 0
    assert out["roofline"]["config1_ms_per_step"] > 0 and "materialize" in out["roofline"]
    assert "configs" in out["extra_keys"] and "configs" not in out

    import copy
    fat = copy.deepcopy(full)
    fat["config"]["workload"] = "BASELINE configs[2]: " + "x" * 20000
    fat["roofline"]["measured_valu_fma_rate_note"] = "y" * 50000
    fat["roofline"]["nan"] = float("nan")
    fat["roofline"]["inf"] = float("inf")
    fat["roofline"]["np"] = np.float64(1.5)
    for i in range(300):                          # a roofline that somebody keeps adding promoted numbers to
        fat["roofline"]["promoted_%d" % i] = {"ms": 1.0 * i, "note": "z" * 100}
    fat["cpu_baseline"]["sample"] = "s" * 9000
    check(fat)

    lean = {k: full[k] for k in bench.HEADLINE_KEYS if k in full}
    lean["config"] = {"workload": full["config"]["workload"]}
    lean["roofline"] = {k: full["roofline"][k] for k in ("bound", "achieved", "peak", "unit", "frac", "traffic")}
    check(lean)


def test_import_celerite_resolves_to_this_build():
    """The north star's "keeping the existing ... API as a drop-in": ``import celerite`` (the reference's package name,
    celerite/__init__.py:20-33) is this build -- package, public names and submodules."""
    import importlib

    import celerite
    import celerite.modeling
    import celerite.solver
    import celerite.terms
    from celerite import GP as GP2, terms as terms2
    from celerite.modeling import Model
    from celerite.solver import CholeskySolver

    assert celerite.GP is GP is GP2 and celerite.terms is terms is terms2 is celerite_amd.terms
    assert celerite.solver is solver and celerite.modeling is modeling and Model is modeling.Model
    assert CholeskySolver is solver.CholeskySolver is celerite.CholeskySolver
    assert importlib.import_module("celerite.terms") is terms
    for name in ("terms", "solver", "modeling", "GP", "CholeskySolver", "__library_version__", "__version__"):
        assert hasattr(celerite, name), name
    assert celerite.__library_version__ == "0.3.0"
    k = celerite.terms.RealTerm(log_a=0.1, log_c=0.2) + celerite.terms.SHOTerm(log_S0=0.1, log_Q=1.0, log_omega0=0.3)
    assert celerite.GP(k).kernel is k


def test_options_table_and_the_split_of_the_headers():
    """``clr_set_option`` replaces round 5's 19 direct ``getenv`` reads: one table per process, environment variables
    honoured only under CLR_ALLOW_ENV=1; the diagnostics live in include/celerite_hip_debug.h, not in the boundary."""
    assert os.environ.get("CLR_ALLOW_ENV") != "1"
    os.environ["CLR_GRAD_SEQUENTIAL"] = "1"            # a stray variable must NOT reach the library
    try:
        assert batch.get_option("CLR_GRAD_SEQUENTIAL") is None
        batch.set_option("CLR_GRAD_SEQUENTIAL", "1")
        assert batch.get_option("CLR_GRAD_SEQUENTIAL") == "1"
        with batch.option("CLR_GRAD_SEQUENTIAL", 0):
            assert batch.get_option("CLR_GRAD_SEQUENTIAL") == "0"
        assert batch.get_option("CLR_GRAD_SEQUENTIAL") == "1"
        batch.set_option("CLR_GRAD_SEQUENTIAL", None)
        assert batch.get_option("CLR_GRAD_SEQUENTIAL") is None
        with pytest.raises(RuntimeError):
            batch.set_option("PATH", "x")
    finally:
        del os.environ["CLR_GRAD_SEQUENTIAL"]
        batch.set_option("CLR_GRAD_SEQUENTIAL", None)
    boundary = open(os.path.join(ROOT, "include", "celerite_hip.h")).read()
    debug = open(os.path.join(ROOT, "include", "celerite_hip_debug.h")).read()
    for name in ("clr_batch_debug_get_starts", "clr_batch_debug_compose_check", "clr_batch_debug_cu_census",
                 "clr_batch_fp32_probe", "clr_device_measure_fp64"):
        assert name + "(" in debug and name + "(" not in boundary, name
    for dirpath, _, files in os.walk(os.path.join(ROOT, "celerite_amd", "csrc")):
        for f in files:
            text = open(os.path.join(dirpath, f)).read()
            if f != "api_misc.hip":
                assert "getenv(" not in text.replace("own getenv", ""), f
