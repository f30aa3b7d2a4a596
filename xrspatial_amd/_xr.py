"""xarray, or a minimal stand-in when it is not installed.

The public functions keep the reference's contract -- xarray.DataArray in,
xarray.DataArray out, Dataset in -> Dataset out.  xarray is not installable in
the build / GPU containers of this project, so a small DataArray / Dataset
with the handful of members this package (and its tests) touch is provided:
data / values / dims / coords / attrs / name / shape / ndim / dtype,
`obj[dim]` coordinate access and assignment, `.equals`.  When real xarray is
importable it is used and this stand-in is dead code.
"""
from __future__ import annotations

import numpy as np

try:                                    # pragma: no cover - not available in this environment
    import xarray as _xarray
    DataArray = _xarray.DataArray
    Dataset = _xarray.Dataset
    HAVE_XARRAY = True
except ImportError:
    _xarray = None
    HAVE_XARRAY = False

    def _host(x):
        return x.get() if hasattr(x, "get") and not isinstance(x, np.ndarray) else np.asarray(x)

    class DataArray:                    # noqa: D101  (minimal xarray.DataArray stand-in)
        def __init__(self, data=None, coords=None, dims=None, name=None, attrs=None):
            if isinstance(data, DataArray):
                coords = data.coords if coords is None else coords
                dims = data.dims if dims is None else dims
                name = data.name if name is None else name
                attrs = data.attrs if attrs is None else attrs
                data = data.data
            if not (hasattr(data, "shape") and hasattr(data, "dtype")):
                data = np.asarray(data)
            self.data = data
            if dims is None:
                dims = tuple(f"dim_{i}" for i in range(len(data.shape)))
            if isinstance(dims, str):
                dims = (dims,)
            self.dims = tuple(dims)
            if len(self.dims) != len(data.shape):
                raise ValueError(f"different number of dimensions on data and dims: "
                                 f"{len(data.shape)} vs {len(self.dims)}")
            self.name = name
            self.attrs = dict(attrs) if attrs else {}
            self.coords = {}
            if coords:
                items = coords.items() if hasattr(coords, "items") else coords
                for k, v in items:
                    self[k] = v

        # -- basic properties --------------------------------------------------
        shape = property(lambda self: tuple(self.data.shape))
        ndim = property(lambda self: len(self.data.shape))
        dtype = property(lambda self: self.data.dtype)
        size = property(lambda self: int(np.prod(self.data.shape, dtype=np.int64)))
        values = property(lambda self: _host(self.data))

        def __array__(self, dtype=None, copy=None):
            a = _host(self.data)
            return a if dtype is None else a.astype(dtype)

        def __getitem__(self, key):
            if isinstance(key, str):
                if key in self.coords:
                    return self.coords[key]
                if key in self.dims:      # default integer index, like xarray
                    n = self.shape[self.dims.index(key)]
                    return DataArray(np.arange(n), dims=(key,), name=key)
                raise KeyError(key)
            sub = _host(self.data)[key]
            if np.ndim(sub) == 0:
                return DataArray(np.asarray(sub), dims=())
            return DataArray(sub)

        def __setitem__(self, key, value):
            if not isinstance(key, str):
                raise TypeError("only coordinate assignment is supported by the xarray stand-in")
            if isinstance(value, DataArray):
                value = DataArray(value.data, dims=value.dims, name=key, attrs=value.attrs)
            else:
                arr = np.asarray(value)
                value = DataArray(arr, dims=(key,) if arr.ndim == 1 else None, name=key)
            self.coords[key] = value

        def min(self):
            return DataArray(np.asarray(np.min(_host(self.data))), dims=())

        def max(self):
            return DataArray(np.asarray(np.max(_host(self.data))), dims=())

        def mean(self):
            return DataArray(np.asarray(np.nanmean(_host(self.data))), dims=())

        def item(self):
            return _host(self.data).item()

        def __float__(self):
            return float(self.item())

        def __gt__(self, other):
            return _host(self.data) > other

        def __lt__(self, other):
            return _host(self.data) < other

        def copy(self, deep=True):
            d = _host(self.data).copy() if deep and isinstance(self.data, np.ndarray) else self.data
            return DataArray(d, coords=self.coords, dims=self.dims, name=self.name, attrs=dict(self.attrs))

        def equals(self, other):
            if not isinstance(other, DataArray) or self.shape != other.shape or self.dims != other.dims:
                return False
            if not np.array_equal(_host(self.data), _host(other.data), equal_nan=True):
                return False
            if set(self.coords) != set(other.coords):
                return False
            return all(np.array_equal(_host(self.coords[k].data), _host(other.coords[k].data))
                       for k in self.coords)

        def to_dataset(self, dim=None):
            if dim is None:                     # xarray: a one-variable Dataset named after the array
                if self.name is None:
                    raise ValueError("unable to convert unnamed DataArray to a Dataset without providing an explicit name")
                return Dataset({self.name: self}, attrs=self.attrs)
            ax = self.dims.index(dim)
            labels = _host(self.coords[dim].data) if dim in self.coords else np.arange(self.shape[ax])
            rest = tuple(d for d in self.dims if d != dim)
            out = {}
            for i, lab in enumerate(labels):
                out[lab.item() if hasattr(lab, "item") else lab] = DataArray(
                    np.take(_host(self.data), i, axis=ax), dims=rest, attrs=self.attrs)
            return Dataset(out, attrs=self.attrs)

        def __repr__(self):
            return f"<DataArray {self.name!r} {dict(zip(self.dims, self.shape))}>\n{self.data!r}"

    class Dataset:                      # noqa: D101  (minimal xarray.Dataset stand-in)
        def __init__(self, data_vars=None, coords=None, attrs=None):
            self.data_vars = {}
            self.attrs = dict(attrs) if attrs else {}
            for k, v in (data_vars or {}).items():
                self[k] = v

        def __getitem__(self, key):
            return self.data_vars[key]

        def __setitem__(self, key, value):
            if not isinstance(value, DataArray):
                value = DataArray(value)
            if value.name is None:
                value.name = key
            self.data_vars[key] = value

        def __contains__(self, key):
            return key in self.data_vars

        def __iter__(self):
            return iter(self.data_vars)

        def __repr__(self):
            return f"<Dataset {list(self.data_vars)}>"
