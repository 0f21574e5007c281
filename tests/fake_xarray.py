"""A minimal stand-in for the ``xarray`` module (TEST INFRASTRUCTURE ONLY).

xarray and ArviZ are absent from the build container and from the GPU box, so the ``isinstance(x, xr.DataArray)`` branches of
``pyloo_amd`` (utils.stack_samples / wrap_obs, base.compute_importance_weights, e_loo._sample_last) would never execute in
this suite.  The tests patch ``pyloo_amd.<module>.xr`` with this module: a ``DataArray`` with just the behaviour those branches
rely on -- ``dims / coords / values / shape / sizes / name``, ``stack(new=(a, b))`` (the stacked dim goes LAST, first name
major, like xarray), ``transpose`` with ``...``, ``rename`` -- so that the branch logic (dim handling, coordinate carry-over,
output wrapping) is exercised.  Where real xarray is installed the same tests run against it as well."""

import numpy as np


class DataArray:
    def __init__(self, data, dims=None, coords=None, name=None):
        self.values = np.asarray(data)
        dims = tuple(dims) if dims is not None else tuple(f"dim_{i}" for i in range(self.values.ndim))
        if len(dims) != self.values.ndim:
            raise ValueError(f"{len(dims)} dims for a {self.values.ndim}-d array")
        self.dims = dims
        coords = dict(coords or {})
        self.coords = {k: np.asarray(getattr(v, "values", v)) for k, v in coords.items() if k in dims}
        self.name = name

    shape = property(lambda self: self.values.shape)
    ndim = property(lambda self: self.values.ndim)
    dtype = property(lambda self: self.values.dtype)
    sizes = property(lambda self: dict(zip(self.dims, self.values.shape)))

    def transpose(self, *dims):
        dims = list(dims)
        if Ellipsis in dims:
            i = dims.index(Ellipsis)
            named = [d for d in dims if d is not Ellipsis]
            dims = dims[:i] + [d for d in self.dims if d not in named] + dims[i + 1:]
        if sorted(dims) != sorted(self.dims):
            raise ValueError(f"{dims} is not a permutation of {self.dims}")
        order = [self.dims.index(d) for d in dims]
        return DataArray(self.values.transpose(order), dims, self.coords, self.name)

    def stack(self, **new):
        (name, parts), = new.items()
        keep = [d for d in self.dims if d not in parts]
        moved = self.transpose(*keep, *parts)
        shape = [moved.sizes[d] for d in keep] + [int(np.prod([moved.sizes[d] for d in parts]))]
        return DataArray(moved.values.reshape(shape), keep + [name], {d: self.coords[d] for d in keep if d in self.coords}, self.name)

    def rename(self, mapping):
        dims = [mapping.get(d, d) for d in self.dims]
        return DataArray(self.values, dims, {mapping.get(k, k): v for k, v in self.coords.items()}, self.name)

    def __array__(self, dtype=None, copy=None):
        return self.values if dtype is None else self.values.astype(dtype)

    def __array_ufunc__(self, ufunc, method, *inputs, **kw):  # np.log(weights) of e_loo.py:202-203
        arrays = [getattr(a, "values", a) for a in inputs]
        return DataArray(getattr(ufunc, method)(*arrays, **kw), self.dims, self.coords, self.name)
