# -*- coding: utf-8 -*-
"""``import celerite`` resolves to the MI355X build (``celerite_amd``): the reference's package name, its public names
(celerite/__init__.py:20-33 of the reference) and its submodules (``celerite.terms``, ``celerite.modeling``,
``celerite.solver``), so that user code written against the reference runs unchanged.  Put this checkout on
``sys.path`` INSTEAD of the reference -- the two cannot be imported side by side under one name."""
import sys as _sys

import celerite_amd as _impl
from celerite_amd import *  # noqa: F401,F403

__all__ = list(getattr(_impl, "__all__", []))
for _name in ("__version__", "__library_version__", "GP", "CholeskySolver", "Model", "ConstantModel"):
    if hasattr(_impl, _name):
        globals()[_name] = getattr(_impl, _name)
for _name in ("terms", "modeling", "solver", "celerite", "batch"):
    _mod = getattr(_impl, _name, None)
    if _mod is not None:
        globals()[_name] = _mod
        _sys.modules[__name__ + "." + _name] = _mod
del _name, _mod
