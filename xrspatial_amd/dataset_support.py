"""Dataset adapters for the DataArray-level public functions.

Behavioural contract (reference: xrspatial/dataset_support.py:11-80, tests
test_dataset_support.py:74-211):
  * `supports_dataset` -- `f(ds, ...)` runs `f` once per data variable, naming each
    result after its variable when `f` accepts `name=`, and returns a Dataset that
    carries the input Dataset's attrs;
  * `supports_dataset_bands(nir='nir_agg', ...)` -- `f(ds, nir='B8', red='B4', ...)`
    resolves the band keywords to variables of `ds`; a missing keyword is a
    TypeError, an unknown variable a ValueError; all other keywords pass through.
"""
from __future__ import annotations

import functools
import inspect

from ._xr import Dataset


def supports_dataset(func):
    takes_name = 'name' in inspect.signature(func).parameters

    def per_variable(ds, args, kwargs):
        out = {}
        for var in ds.data_vars:
            extra = {'name': var} if takes_name else {}
            out[var] = func(ds[var], *args, **{**kwargs, **extra})
        return Dataset(out, attrs=ds.attrs)

    @functools.wraps(func)
    def wrapper(agg, *args, **kwargs):
        if isinstance(agg, Dataset):
            return per_variable(agg, args, kwargs)
        return func(agg, *args, **kwargs)

    return wrapper


def supports_dataset_bands(**alias_to_param):
    def resolve(ds, kwargs):
        call_kwargs = {k: v for k, v in kwargs.items() if k not in alias_to_param}
        for alias, param in alias_to_param.items():
            if alias not in kwargs:
                raise TypeError(f"'{alias}' keyword required when passing a Dataset")
            var = kwargs[alias]
            if var not in ds.data_vars:
                raise ValueError(f"'{var}' not in Dataset. Available: {list(ds.data_vars)}")
            call_kwargs[param] = ds[var]
        return call_kwargs

    def decorator(func):
        @functools.wraps(func)
        def wrapper(*args, **kwargs):
            if args and isinstance(args[0], Dataset):
                return func(**resolve(args[0], kwargs))
            return func(*args, **kwargs)
        return wrapper

    return decorator
