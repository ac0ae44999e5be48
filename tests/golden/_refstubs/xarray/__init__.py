"""Minimal stand-in for xarray, used ONLY in the build container (which has no xarray and no network) so that the
reference package can be imported as a fixture generator (tests/golden/make_golden.py) and so that its own `Model.run`
can nest the results an rtsolver plugin hands back (tests/test_reference_binding.py).  Holds values / coords / attrs and
stacks along a new dimension like `xr.concat(objs, pd.Index, join="outer")`; no selection logic.  Not shipped, never
imported by the product."""
import numpy as np


class DataArray:
    def __init__(self, data=None, coords=None, dims=None, name=None, attrs=None):
        self.values = np.asarray(data)
        self.name = name
        self.attrs = dict(attrs or {})
        if coords is not None and not isinstance(coords, dict):
            self.coords = [(c[0], np.asarray(c[1])) if isinstance(c, tuple) else ("dim_%d" % i, np.asarray(list(c)))
                           for i, c in enumerate(coords)]
            self.dims = tuple(c[0] for c in self.coords)
        else:
            self.coords = coords
            self.dims = tuple(dims or ())


def concat(objs, dim, join="outer", **kwargs):
    """Stack DataArrays along a new leading dimension named after the pandas Index `dim`; ragged trailing dimensions
    are padded with NaN on the union of their coordinates (what join="outer" does for integer range coordinates)."""
    objs = list(objs)
    name = getattr(dim, "name", None) or "concat_dim"
    labels = np.asarray(list(dim))
    nd = objs[0].values.ndim
    shape = tuple(max(o.values.shape[k] for o in objs) for k in range(nd))
    ragged = any(o.values.shape != shape for o in objs)
    dtype = np.result_type(*[o.values.dtype for o in objs], np.float64 if ragged else objs[0].values.dtype)
    out = np.full((len(objs),) + shape, np.nan, dtype=dtype)
    for k, o in enumerate(objs):
        out[(k,) + tuple(slice(0, n) for n in o.values.shape)] = o.values
    longest = [max((o.coords[k] for o in objs), key=lambda c: len(c[1])) for k in range(nd)]
    return DataArray(out, coords=[(name, labels)] + longest, name=objs[0].name, attrs=objs[0].attrs)


def open_dataarray(*a, **k):
    raise NotImplementedError("stub")
