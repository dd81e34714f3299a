# -*- coding: utf-8 -*-
"""celerite_amd: the celerite semiseparable-Cholesky hot path on AMD MI355X.

Drop-in for the public surface of the reference package ``celerite``
(celerite/__init__.py:20-33): ``GP``, ``CholeskySolver``, ``terms``,
``solver``, ``modeling`` and ``__library_version__``, plus the new batched
entry point :mod:`celerite_amd.batch`.

``import celerite_amd as celerite`` is the intended switch for a user of the
reference.  All factorisations and solves run as hand-written HIP kernels for
gfx950 through the C ABI in ``include/celerite_hip.h``; there is no CPU
implementation in the product (calls raise without an MI355X).
"""
__version__ = "0.4.2+mi355x.r1"

__bibtex__ = """
@article{celerite,
    author = {{Foreman-Mackey}, D. and {Agol}, E. and {Angus}, R. and
              {Ambikasaran}, S.},
     title = {Fast and scalable Gaussian process modeling
              with applications to astronomical time series},
      year = {2017},
   journal = {AJ},
    volume = {154},
     pages = {220},
       doi = {10.3847/1538-3881/aa9332},
       url = {https://arxiv.org/abs/1703.09710}
}
"""

__all__ = [
    "terms",
    "solver",
    "modeling",
    "batch",
    "GP",
    "CholeskySolver",
    "__library_version__",
]

try:
    from . import solver
except ImportError as exc:  # pragma: no cover - build problem, make it loud
    raise ImportError(
        "celerite_amd.solver (the compiled HIP extension) is missing or failed to "
        "load: run `make` at the repository root (or `python -c 'import "
        "__graft_entry__ as g; g.build()'`). There is no pure-Python fallback. "
        "Original error: {0}".format(exc)
    )
from . import terms, modeling, batch
from .celerite import GP
from .solver import CholeskySolver

__library_version__ = solver.get_library_version()
