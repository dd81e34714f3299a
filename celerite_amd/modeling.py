# -*- coding: utf-8 -*-
"""Parameter-vector protocol shared by kernels, mean models and the GP.

Own implementation of the interface the reference exposes in
``celerite/modeling.py`` (``Model`` :11-312, ``ModelSet`` :315-431,
``ConstantModel`` :434-447): named float parameters, freeze/thaw masks, bounds
used as a flat prior, and a ``dirty`` flag that every mutation raises so the GP
knows to refactorise (``celerite/celerite.py:95-101,160-171``).  No arithmetic
lives here.
"""
from collections import OrderedDict

import numpy as np

__all__ = ["Model", "ModelSet", "ConstantModel"]


class Model(object):
    """A set of named scalar parameters.

    Subclasses list their parameters in ``parameter_names``; values are given
    positionally or by keyword.  ``bounds`` (list of ``(min, max)`` in
    parameter order, or a dict by name; ``None`` = unbounded) feeds
    :meth:`log_prior`.  ``quiet=True`` skips the finite-prior check at
    construction (``tests/test_terms.py:122-139``).
    """

    parameter_names = tuple()

    def __init__(self, *args, **kwargs):
        n = self.full_size
        self.unfrozen_mask = np.ones(n, dtype=bool)
        self.dirty = True

        bounds = kwargs.pop("bounds", None)
        if bounds is None:
            limits = [(None, None)] * n
        elif hasattr(bounds, "get"):
            limits = [bounds.get(name, (None, None)) for name in self.parameter_names]
        else:
            limits = list(bounds)
        if len(limits) != n:
            raise ValueError("the number of bounds must equal the number of parameters")
        if any(len(pair) != 2 for pair in limits):
            raise ValueError("the bounds for each parameter must have the format: '(min, max)'")
        self.parameter_bounds = limits

        quiet = kwargs.pop("quiet", False)

        if args:
            if len(args) != n:
                raise ValueError("expected {0} arguments but got {1}".format(n, len(args)))
            if kwargs:
                raise ValueError("parameters must be fully specified by arguments "
                                 "or keyword arguments, not both")
            values = args
        else:
            values = []
            for name in self.parameter_names:
                if kwargs.get(name) is None:
                    raise ValueError("missing parameter '{0}'".format(name))
                values.append(kwargs.pop(name))
            if kwargs:
                raise ValueError("unrecognized parameter(s) '{0}'".format(list(kwargs)))
        self.parameter_vector = values

        if not quiet and not np.isfinite(self.log_prior()):
            raise ValueError("non-finite log prior value")

    # -- to be provided by concrete models ---------------------------------
    def get_value(self, *args, **kwargs):
        raise NotImplementedError("overloaded by subclasses")

    def compute_gradient(self, *args, **kwargs):
        raise NotImplementedError("overloaded by subclasses")

    def get_gradient(self, *args, **kwargs):
        include_frozen = kwargs.pop("include_frozen", False)
        grad = self.compute_gradient(*args, **kwargs)
        return grad if include_frozen else grad[self.unfrozen_mask]

    # -- sizes ---------------------------------------------------------------
    @property
    def full_size(self):
        """Number of parameters, frozen ones included."""
        return len(self.parameter_names)

    @property
    def vector_size(self):
        """Number of thawed parameters."""
        return self.unfrozen_mask.sum()

    def __len__(self):
        return self.vector_size

    # -- the full vector -------------------------------------------------------
    @property
    def parameter_vector(self):
        return np.array([getattr(self, name) for name in self.parameter_names])

    @parameter_vector.setter
    def parameter_vector(self, values):
        if len(values) != self.full_size:
            raise ValueError("dimension mismatch")
        for name, value in zip(self.parameter_names, values):
            setattr(self, name, float(value))
        self.dirty = True

    # -- filtered views ----------------------------------------------------------
    def _select(self, seq, include_frozen):
        if include_frozen:
            return seq
        return [item for item, keep in zip(seq, self.unfrozen_mask) if keep]

    def get_parameter_names(self, include_frozen=False):
        if include_frozen:
            return self.parameter_names
        return tuple(self._select(self.parameter_names, False))

    def get_parameter_bounds(self, include_frozen=False):
        if include_frozen:
            return self.parameter_bounds
        return list(self._select(self.parameter_bounds, False))

    def get_parameter_vector(self, include_frozen=False):
        full = self.parameter_vector
        return full if include_frozen else full[self.unfrozen_mask]

    def get_parameter_dict(self, include_frozen=False):
        return OrderedDict(zip(self.get_parameter_names(include_frozen=include_frozen),
                               self.get_parameter_vector(include_frozen=include_frozen)))

    def set_parameter_vector(self, vector, include_frozen=False):
        full = self.parameter_vector
        if include_frozen:
            full[:] = vector
        else:
            full[self.unfrozen_mask] = vector
        self.parameter_vector = full
        self.dirty = True

    # -- by name / index ------------------------------------------------------------
    def _index_of(self, name):
        return self.get_parameter_names(include_frozen=True).index(name)

    def get_parameter(self, name):
        return self.get_parameter_vector(include_frozen=True)[self._index_of(name)]

    def set_parameter(self, name, value):
        full = self.get_parameter_vector(include_frozen=True)
        full[self._index_of(name)] = value
        self.set_parameter_vector(full, include_frozen=True)

    def _resolve(self, key):
        try:
            idx = int(key)
        except (TypeError, ValueError):
            return key
        return self.get_parameter_names()[idx]

    def __getitem__(self, key):
        return self.get_parameter(self._resolve(key))

    def __setitem__(self, key, value):
        return self.set_parameter(self._resolve(key), value)

    # -- freezing ----------------------------------------------------------------------
    def freeze_parameter(self, name):
        self.unfrozen_mask[self._index_of(name)] = False

    def thaw_parameter(self, name):
        self.unfrozen_mask[self._index_of(name)] = True

    def freeze_all_parameters(self):
        self.unfrozen_mask[:] = False

    def thaw_all_parameters(self):
        self.unfrozen_mask[:] = True

    # -- prior ---------------------------------------------------------------------------
    def log_prior(self):
        """0 inside the bounds, -inf outside."""
        for value, (lo, hi) in zip(self.parameter_vector, self.parameter_bounds):
            if (lo is not None and value < lo) or (hi is not None and value > hi):
                return -np.inf
        return 0.0


class ModelSet(Model):
    """An ordered collection of named sub-models behaving as one Model.

    Parameter names are ``"<model>:<parameter>"``; every vector-valued property
    is the concatenation over the sub-models in insertion order.
    """

    def __init__(self, models):
        self.models = OrderedDict(models)

    def __getattr__(self, name):
        members = self.__dict__.get("models")
        if members is not None and name in members:
            return members[name]
        raise AttributeError(name)

    def _each(self):
        return self.models.values()

    @property
    def dirty(self):
        return any(m.dirty for m in self._each())

    @dirty.setter
    def dirty(self, flag):
        for m in self._each():
            m.dirty = flag

    @property
    def full_size(self):
        return sum(m.full_size for m in self._each())

    @property
    def vector_size(self):
        return sum(m.vector_size for m in self._each())

    @property
    def unfrozen_mask(self):
        return np.concatenate([m.unfrozen_mask for m in self._each()])

    @property
    def parameter_vector(self):
        return np.concatenate([m.parameter_vector for m in self._each()])

    @parameter_vector.setter
    def parameter_vector(self, values):
        start = 0
        for m in self._each():
            stop = start + m.full_size
            m.parameter_vector = values[start:stop]
            start = stop

    @property
    def parameter_names(self):
        return tuple("{0}:{1}".format(prefix, name)
                     for prefix, m in self.models.items()
                     for name in m.parameter_names)

    @property
    def parameter_bounds(self):
        return [pair for m in self._each() for pair in m.parameter_bounds]

    def _delegate(self, method, name, *args):
        prefix, _, rest = name.partition(":")
        if prefix not in self.models:
            raise ValueError("unrecognized parameter '{0}'".format(name))
        return getattr(self.models[prefix], method)(rest, *args)

    def freeze_parameter(self, name):
        self._delegate("freeze_parameter", name)

    def thaw_parameter(self, name):
        self._delegate("thaw_parameter", name)

    def freeze_all_parameters(self):
        for m in self._each():
            m.freeze_all_parameters()

    def thaw_all_parameters(self):
        for m in self._each():
            m.thaw_all_parameters()

    def get_parameter(self, name):
        return self._delegate("get_parameter", name)

    def set_parameter(self, name, value):
        self.dirty = True
        return self._delegate("set_parameter", name, value)

    def log_prior(self):
        total = 0.0
        for m in self._each():
            total += m.log_prior()
            if not np.isfinite(total):
                return -np.inf
        return total


class ConstantModel(Model):
    """``value`` everywhere (the GP's default mean)."""

    parameter_names = ("value", )

    def get_value(self, x):
        return self.value + np.zeros_like(x)

    def compute_gradient(self, x):
        return np.array([np.ones_like(x)])
