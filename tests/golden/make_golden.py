# -*- coding: utf-8 -*-
"""Regenerates the golden fixtures in this directory FROM THE REFERENCE.

Runs only in the build container (needs /root/reference, read-only); the
fixtures it writes are plain data (inputs + expected outputs) and are what
travels.  Nothing in tests/, bench.py or the package reads /root/reference at
run time.

Fixtures
--------
terms_golden.json
    Coefficient blocks + jitter produced by the REFERENCE's own
    ``celerite/terms.py`` (imported here with a stub standing in for the
    compiled ``celerite.solver`` module, which ``terms.py:18`` only needs for
    three names) for every kernel the reference's tests construct
    (tests/test_celerite.py:311-433, tests/test_terms.py:13-62).
ipynb_golden.npz
    Inputs and outputs of the authors' NumPy prototype of the factorisation
    and solve recurrences, /root/reference/cholesky.ipynb cells 0, 4, 5,
    executed here with a fixed seed and N reduced to 300 (the removed alias
    ``np.complex`` patched to ``complex``).  An implementation of the algorithm
    that is independent of cholesky.h.
"""
import importlib.util
import json
import os
import sys
import types

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def _import_reference_terms():
    pkg = types.ModuleType("celerite")
    pkg.__path__ = [os.path.join(REF, "celerite")]
    sys.modules["celerite"] = pkg
    stub = types.ModuleType("celerite.solver")

    def _unavailable(*a, **k):
        raise RuntimeError("compiled reference solver is not buildable here")

    stub.get_kernel_value = stub.get_psd_value = stub.check_coefficients = _unavailable
    sys.modules["celerite.solver"] = stub
    for name in ("modeling", "terms"):
        spec = importlib.util.spec_from_file_location(
            "celerite." + name, os.path.join(REF, "celerite", name + ".py"))
        mod = importlib.util.module_from_spec(spec)
        sys.modules["celerite." + name] = mod
        spec.loader.exec_module(mod)
    return sys.modules["celerite.terms"]


def make_terms_golden():
    terms = _import_reference_terms()
    T = terms
    cases = {
        "real": ("RealTerm(log_a=0.1, log_c=0.5)", T.RealTerm(log_a=0.1, log_c=0.5)),
        "real+real": ("RealTerm(log_a=0.1, log_c=0.5) + RealTerm(log_a=-0.1, log_c=0.7)",
                      T.RealTerm(log_a=0.1, log_c=0.5) + T.RealTerm(log_a=-0.1, log_c=0.7)),
        "complex3": ("ComplexTerm(log_a=0.1, log_c=0.5, log_d=0.1)",
                     T.ComplexTerm(log_a=0.1, log_c=0.5, log_d=0.1)),
        "complex4": ("ComplexTerm(log_a=0.1, log_b=-0.2, log_c=0.5, log_d=0.1)",
                     T.ComplexTerm(log_a=0.1, log_b=-0.2, log_c=0.5, log_d=0.1)),
        "jitter": ("JitterTerm(log_sigma=0.1)", T.JitterTerm(log_sigma=0.1)),
        "sho_lowQ+jitter": ("SHOTerm(log_S0=0.1, log_Q=-1, log_omega0=0.5) + JitterTerm(log_sigma=0.1)",
                            T.SHOTerm(log_S0=0.1, log_Q=-1, log_omega0=0.5) + T.JitterTerm(log_sigma=0.1)),
        "sho_lowQ": ("SHOTerm(log_S0=0.1, log_Q=-1, log_omega0=0.5)",
                     T.SHOTerm(log_S0=0.1, log_Q=-1, log_omega0=0.5)),
        "sho_highQ": ("SHOTerm(log_S0=0.1, log_Q=1.0, log_omega0=0.5)",
                      T.SHOTerm(log_S0=0.1, log_Q=1.0, log_omega0=0.5)),
        "sho+real": ("SHOTerm(log_S0=0.1, log_Q=1.0, log_omega0=0.5) + RealTerm(log_a=0.1, log_c=0.4)",
                     T.SHOTerm(log_S0=0.1, log_Q=1.0, log_omega0=0.5) + T.RealTerm(log_a=0.1, log_c=0.4)),
        "sho*real": ("SHOTerm(log_S0=0.1, log_Q=1.0, log_omega0=0.5) * RealTerm(log_a=0.1, log_c=0.4)",
                     T.SHOTerm(log_S0=0.1, log_Q=1.0, log_omega0=0.5) * T.RealTerm(log_a=0.1, log_c=0.4)),
        "real+sho (config 1)": ("RealTerm(0.1, 0.5) + SHOTerm(log_S0=0.1, log_Q=1.0, log_omega0=0.5)",
                                T.RealTerm(0.1, 0.5) + T.SHOTerm(log_S0=0.1, log_Q=1.0, log_omega0=0.5)),
        "matern32": ("Matern32Term(log_sigma=0.3, log_rho=-0.4)",
                     T.Matern32Term(log_sigma=0.3, log_rho=-0.4)),
        "matern32_eps": ("Matern32Term(log_sigma=0.3, log_rho=-0.4, eps=0.1)",
                         T.Matern32Term(log_sigma=0.3, log_rho=-0.4, eps=0.1)),
        "complex*complex": ("ComplexTerm(0.2, -3.0, 0.5, 0.01) * ComplexTerm(0.6, 0.7, 1.0)",
                            T.ComplexTerm(0.2, -3.0, 0.5, 0.01) * T.ComplexTerm(0.6, 0.7, 1.0)),
        "(real+complex)*sho": ("(RealTerm(log_a=0.1, log_c=0.5) + ComplexTerm(0.2, -3.0, 0.5, 0.01)) * SHOTerm(1.0, 0.2, 3.0)",
                               (T.RealTerm(log_a=0.1, log_c=0.5) + T.ComplexTerm(0.2, -3.0, 0.5, 0.01)) * T.SHOTerm(1.0, 0.2, 3.0)),
        "bench width 8": ("RealTerm(1.0, 0.1) + RealTerm(1.0, 0.1) + 3 x ComplexTerm(0.1, 2.0, 1.6)",
                          T.RealTerm(1.0, 0.1) + T.RealTerm(1.0, 0.1) + T.ComplexTerm(0.1, 2.0, 1.6)
                          + T.ComplexTerm(0.1, 2.0, 1.6) + T.ComplexTerm(0.1, 2.0, 1.6)),
    }
    # the term-by-term kernel of tests/test_celerite.py:346-353
    kernel = T.RealTerm(0.1, 0.5)
    termlist = [(0.1 + 10. / j, 0.5 + 10. / j) for j in range(1, 4)]
    termlist += [(1.0 + 10. / j, 0.01 + 10. / j, 0.5, 0.01) for j in range(1, 10)]
    termlist += [(0.6, 0.7, 1.0), (0.3, 0.05, 0.5, 0.6)]
    for term in termlist:
        kernel += T.ComplexTerm(*term) if len(term) > 2 else T.RealTerm(*term)
    cases["test_log_likelihood full kernel (width 26)"] = ("see tests/test_celerite.py:346-353", kernel)

    out = {}
    for key, (expr, k) in cases.items():
        out[key] = dict(
            expr=expr,
            parameter_names=list(k.get_parameter_names(include_frozen=True)),
            parameter_vector=[float(v) for v in k.get_parameter_vector(include_frozen=True)],
            coefficients=[[float(x) for x in block] for block in k.coefficients],
            jitter=float(k.jitter),
            repr=repr(k),
        )
    with open(os.path.join(HERE, "terms_golden.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    return out


def make_ipynb_golden(N=300, seed=1234):
    nb = json.load(open(os.path.join(REF, "cholesky.ipynb")))
    cells = ["".join(c["source"]) for c in nb["cells"] if c["cell_type"] == "code"]
    env = {}
    np.random.seed(seed)
    src0 = cells[0].replace("N = 1000", "N = %d" % N)
    exec(compile(src0, "cholesky.ipynb[0]", "exec"), env)
    y0 = np.array(env["y0"])
    Kc = np.array(env["K"].real)
    exec(compile(cells[4].replace("np.complex)", "complex)"), "cholesky.ipynb[4]", "exec"), env)
    D = np.array(env["D"])
    X1 = np.array(env["X1"])
    X2 = np.array(env["X2"])
    exec(compile(cells[5], "cholesky.ipynb[5]", "exec"), env)
    z = np.array(env["z"])
    np.savez_compressed(
        os.path.join(HERE, "ipynb_golden.npz"),
        a=env["a"], b=env["b"], c=env["c"], d=env["d"], t=env["t"],
        diag_full=env["diag"],  # the FULL diagonal of K (includes sum(a))
        y0=y0, D=D, X1=X1, X2=X2, solve=z,
        logdet_dense=np.linalg.slogdet(np.array(env["K2"]))[1],
        logdet_chol=np.sum(np.log(D)),
        K_maxdiff=np.max(np.abs(Kc - env["K2"])),
    )


if __name__ == "__main__":
    make_terms_golden()
    make_ipynb_golden()
    print("wrote", sorted(f for f in os.listdir(HERE) if not f.endswith(".py")))
