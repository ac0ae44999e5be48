"""Minimal stand-in for xarray, used ONLY by tests/golden/make_golden.py in the build container so that the
reference's rtsolver can hand back its ndarray.  Holds values/coords/attrs; no selection logic."""
import numpy as np


class DataArray:
    def __init__(self, data=None, coords=None, dims=None, name=None, attrs=None):
        self.values = np.asarray(data)
        self.coords = coords
        self.name = name
        self.attrs = dict(attrs or {})
        if coords is not None and not isinstance(coords, dict):
            self.dims = tuple(c[0] if isinstance(c, tuple) else "dim_%d" % i for i, c in enumerate(coords))
        else:
            self.dims = tuple(dims or ())


def concat(*a, **k):
    raise NotImplementedError("stub")


def open_dataarray(*a, **k):
    raise NotImplementedError("stub")
