"""Run-time defaults (``rcParams``) with the same keys and validation rules the hot path reads
in the reference (pyloo/rcparams.py:30-34; ``loo`` uses ``stats.ic_pointwise`` and
``stats.ic_scale``: loo.py:181,193)."""

from collections.abc import MutableMapping

_SCALES = ("deviance", "log", "negative_log")


def _check_bool(v):
    if isinstance(v, bool):
        return v
    raise ValueError(f"Value must be True or False, not {v}")


def _check_scale(v):
    if isinstance(v, str) and v.lower() in _SCALES:
        return v.lower()
    raise ValueError(f"Scale must be one of {set(_SCALES)}, not {v}")


def _check_backend(v):
    if isinstance(v, str) and v.lower() == "matplotlib":
        return v.lower()
    raise ValueError(f"Backend must be one of {{'matplotlib'}}, not {v}")


_DEFAULTS = {
    "stats.ic_pointwise": (False, _check_bool),
    "stats.ic_scale": ("log", _check_scale),
    "plot.backend": ("matplotlib", _check_backend),
}


class RcParams(MutableMapping):
    """Validated mapping; keys cannot be added or removed (rcparams.py:37-118)."""

    def __init__(self, *args, **kwargs):
        self._store = {k: v for k, (v, _) in _DEFAULTS.items()}
        self.update(*args, **kwargs)

    def __setitem__(self, key, val):
        if key not in _DEFAULTS:
            raise KeyError(
                f"{key} is not a valid rc parameter (see rcParams.keys() for a list of valid parameters)"
            )
        try:
            self._store[key] = _DEFAULTS[key][1](val)
        except ValueError as err:
            raise ValueError(f"Key {key}: {err}") from err

    def __getitem__(self, key):
        return self._store[key]

    # the mapping is frozen: every way of removing or implicitly adding a key raises (rcparams.py:75-105)
    def _frozen(msg):  # noqa: N805 - helper evaluated in the class body
        def method(self, *args, **kwargs):
            raise TypeError(msg)

        return method

    __delitem__ = clear = _frozen("RcParams keys cannot be deleted")
    pop = popitem = _frozen("RcParams keys cannot be deleted. Use .get(key) or RcParams[key] to check values")
    setdefault = _frozen("Defaults in RcParams are handled on object initialization. Use pyloo configuration file instead.")
    del _frozen

    def __iter__(self):
        return iter(sorted(self._store))

    def __len__(self):
        return len(self._store)

    def __repr__(self):
        return f"{type(self).__name__}({self._store})"

    def __str__(self):
        return "\n".join(f"{k:<22}: {v}" for k, v in sorted(self._store.items()))

    def copy(self):
        return dict(self._store)


rcParams = RcParams()
