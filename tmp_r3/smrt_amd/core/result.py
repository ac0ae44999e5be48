"""Result / PassiveResult / ActiveResult without xarray: a small labelled n-d array with `sel`, and the accessor
surface of smrt/core/result.py:93-680 (TbV/TbH/Tb(channel=)/sigmaVV.../to_dataframe/other_data) plus
concat_results (:768-817)."""
from collections import OrderedDict

import numpy as np

from ..utils import dB
from .error import SMRTError


class LabeledArray:
    """values + ordered named coordinates; `sel(name=value, ...)` drops the selected dimensions."""

    def __init__(self, values, coords, name=None, attrs=None):
        self.values = np.asarray(values)
        self.coords = OrderedDict((k, np.asarray(v)) for k, v in coords)
        self.name = name
        self.attrs = dict(attrs or {})
        if self.values.shape != tuple(len(v) for v in self.coords.values()):
            raise SMRTError(f"shape {self.values.shape} does not match coords "
                            f"{[(k, len(v)) for k, v in self.coords.items()]}")

    @property
    def dims(self):
        return tuple(self.coords.keys())

    @property
    def shape(self):
        return self.values.shape

    @property
    def size(self):
        return self.values.size

    def __getattr__(self, attr):
        coords = self.__dict__.get("coords", {})
        if attr in coords:
            return coords[attr]
        raise AttributeError(attr)

    def _index(self, dim, value):
        c = self.coords[dim]
        if c.dtype.kind in "fc":
            hit = np.nonzero(np.isclose(c, value, rtol=1e-12, atol=0))[0]
        else:
            hit = np.nonzero(c == value)[0]
        if len(hit) == 0:
            raise KeyError(f"{value!r} not found in dimension '{dim}'")
        return int(hit[0])

    def sel(self, drop=True, **kwargs):
        idx = []
        new_coords = []
        for d in self.dims:
            if d in kwargs and not isinstance(kwargs[d], (list, tuple, np.ndarray)):
                idx.append(self._index(d, kwargs[d]))
            elif d in kwargs:
                ii = [self._index(d, v) for v in kwargs[d]]
                idx.append(ii)
                new_coords.append((d, self.coords[d][ii]))
            else:
                idx.append(slice(None))
                new_coords.append((d, self.coords[d]))
        unknown = set(kwargs) - set(self.dims)
        if unknown:
            raise KeyError(f"unknown dimension(s) {sorted(unknown)}")
        vals = self.values
        for axis in reversed(range(len(idx))):  # apply one axis at a time (no fancy-index broadcasting)
            vals = np.take(vals, idx[axis], axis=axis) if not isinstance(idx[axis], slice) else vals
        return LabeledArray(vals, new_coords, name=self.name, attrs=self.attrs)

    def rename(self, name):
        return LabeledArray(self.values, list(self.coords.items()), name=name, attrs=self.attrs)

    def squeeze(self):
        keep = [(k, v) for k, v in self.coords.items() if len(v) != 1]
        return LabeledArray(self.values.reshape([len(v) for _, v in keep]), keep, name=self.name, attrs=self.attrs)

    def __float__(self):
        return float(self.values.reshape(-1)[0]) if self.size == 1 else float(self.values)

    def __array__(self, dtype=None, copy=None):
        return np.asarray(self.values, dtype=dtype)

    def _binary(self, other, op):
        o = other.values if isinstance(other, LabeledArray) else other
        return LabeledArray(op(self.values, o), list(self.coords.items()), name=self.name, attrs=self.attrs)

    def __getitem__(self, key):
        return self.values[key]

    def __len__(self):
        return len(self.values)

    def __iter__(self):
        return iter(self.values)

    def __sub__(self, o):
        return self._binary(o, np.subtract)

    def __rsub__(self, o):
        return self._binary(o, lambda a, b: np.subtract(b, a))

    def __neg__(self):
        return LabeledArray(-self.values, list(self.coords.items()), name=self.name, attrs=self.attrs)

    def __lt__(self, o):
        return self.values < (o.values if isinstance(o, LabeledArray) else o)

    def __le__(self, o):
        return self.values <= (o.values if isinstance(o, LabeledArray) else o)

    def __gt__(self, o):
        return self.values > (o.values if isinstance(o, LabeledArray) else o)

    def __ge__(self, o):
        return self.values >= (o.values if isinstance(o, LabeledArray) else o)

    def __mul__(self, o):
        return self._binary(o, np.multiply)

    __rmul__ = __mul__

    def __add__(self, o):
        return self._binary(o, np.add)

    __radd__ = __add__

    def __truediv__(self, o):
        return self._binary(o, np.divide)

    def to_dataframe(self, name=None):
        import pandas as pd

        name = name or self.name or "value"
        if not self.dims:
            return pd.DataFrame([float(self.values)], columns=[name])
        index = pd.MultiIndex.from_product(list(self.coords.values()), names=list(self.coords.keys()))
        if len(self.dims) == 1:
            index = pd.Index(list(self.coords.values())[0], name=self.dims[0])
        return pd.DataFrame({name: self.values.reshape(-1)}, index=index)

    def __repr__(self):
        return f"LabeledArray(name={self.name!r}, dims={self.dims}, values=\n{self.values})"


NETCDF_VARIABLE = "__xarray_dataarray_variable__"   # the name xarray gives an unnamed DataArray in DataArray.to_netcdf


def save_labeled_array(arr, filename):
    """Write a LabeledArray as a netCDF-3 file laid out like xarray's DataArray.to_netcdf (one data variable, one
    coordinate variable per dimension, strings as character arrays, attributes on the data variable), so that the
    reference's `open_result` / `xr.open_dataarray` read it.  scipy.io.netcdf_file is the only dependency."""
    from scipy.io import netcdf_file

    with netcdf_file(filename, "w", version=2) as nc:
        for dim, values in arr.coords.items():
            values = np.asarray(values)
            nc.createDimension(dim, len(values))
            if values.dtype.kind in "US":
                width = max(1, max(len(str(v)) for v in values))
                sdim = "string%d" % width
                if sdim not in nc.dimensions:
                    nc.createDimension(sdim, width)
                var = nc.createVariable(dim, "c", (dim, sdim))
                var[:] = np.array([list(str(v).ljust(width, "\0")) for v in values], dtype="S1")
            else:
                var = nc.createVariable(dim, "d" if values.dtype.kind == "f" else "i", (dim,))
                var[:] = values
        data = nc.createVariable(NETCDF_VARIABLE, "d", tuple(arr.coords.keys()))
        data[:] = np.asarray(arr.values, dtype=np.float64)
        for k, v in arr.attrs.items():
            setattr(data, k, v)


def load_labeled_array(filename):
    from scipy.io import netcdf_file

    with netcdf_file(filename, "r", mmap=False) as nc:
        names = [n for n in nc.variables if n not in nc.dimensions]
        if len(names) != 1:
            raise SMRTError(f"'{filename}' does not hold exactly one data variable")
        var = nc.variables[names[0]]
        coords = []
        for dim in var.dimensions:
            c = nc.variables[dim]
            v = c[:].copy()
            if v.dtype.kind == "S" and v.ndim == 2:
                v = np.array([b"".join(row).rstrip(b"\0").decode() for row in v])
            coords.append((dim, v))
        attrs = {k: (v.decode() if isinstance(v, bytes) else v) for k, v in var._attributes.items()}
        return LabeledArray(var[:].copy(), coords, attrs=attrs)


def open_result(filename):
    """Read a result written by `Result.save` (or by the reference's, when it was written in a netCDF-3 format)."""
    data = load_labeled_array(filename)
    mode = data.attrs.get("mode")
    if mode not in ("A", "P"):
        mode = "A" if "theta_inc" in data.coords else "P"
    return (ActiveResult if mode == "A" else PassiveResult)(data)


def _strongsqueeze(x):
    x = x.squeeze()
    return float(x) if x.size == 1 else x


def concat_arrays(arrays, dim_name, dim_values):
    """Stack along a new leading dimension (what xr.concat(..., pd.Index) does at result.py:811)."""
    first = arrays[0]
    for a in arrays[1:]:
        if a.dims != first.dims or any(not np.array_equal(a.coords[d], first.coords[d]) for d in first.dims):
            raise SMRTError("cannot concatenate results with different coordinates")
    vals = np.stack([a.values for a in arrays], axis=0)
    return LabeledArray(vals, [(dim_name, np.asarray(list(dim_values)))] + list(first.coords.items()),
                        name=first.name, attrs=first.attrs)


class Result(object):
    """Contain the results of a/many computations and provide convenience accessors."""

    def __init__(self, radiance, coords=None, channel_map=None, other_data=None, mother_df=None):
        if isinstance(radiance, LabeledArray):
            self.data = radiance
        else:
            self.data = LabeledArray(radiance, coords)
        self.other_data = other_data or {}
        self.mother_df = mother_df
        if not hasattr(self, "mode"):
            raise SMRTError("Result base class is abstract, uses a subclass instead.")
        self.data.attrs["mode"] = self.mode
        self.channel_map = channel_map or dict()

    @property
    def coords(self):
        return self.data.coords

    def __getattr__(self, attr):
        data = self.__dict__.get("data")
        if attr != "data" and data is not None and attr in data.coords:
            return data.coords[attr]
        raise AttributeError(f"'{type(self)}' object has no attribute '{attr}'")

    def save(self, filename, netcdf_engine=None):
        """Save the result to disk as a netCDF file (the data array with its coordinates, like the reference's
        Result.save, smrt/core/result.py:138-147); read it back with `open_result`."""
        save_labeled_array(self.data, filename)

    def sel_data(self, channel=None, **kwargs):
        raise NotImplementedError

    def return_as_dataframe(self, name, channel_axis=None, **kwargs):
        import pandas as pd

        if channel_axis in ("column", "index"):
            if not self.channel_map:
                raise SMRTError("No channel information is given in the result. Unable to index the result by channel.")
            df = pd.concat([self.sel_data(channel=ch, **kwargs).to_dataframe(name=ch) for ch in self.channel_map],
                           axis=1, join="inner")
            if channel_axis == "index":
                df = df.stack()
                if isinstance(df, pd.Series):
                    df = pd.DataFrame(df, columns=[name])
        elif channel_axis is None:
            df = self.sel_data(**kwargs).to_dataframe(name=name)
        else:
            raise SMRTError('channel_axis argument must be None, "column" or "index"')
        if self.mother_df is not None and channel_axis == "column":
            df = df.reset_index(drop=True).join(self.mother_df.reset_index(drop=True))
            df.index = self.mother_df.index
        return df

    def to_series(self, **kwargs):
        return self.return_as_dataframe("out", channel_axis="column", **kwargs).iloc[0]

    def optical_depth(self):
        return (self.other_data["ka"] + self.other_data["ks"]) * self.other_data["thickness"]

    def single_scattering_albedo(self):
        return self.other_data["ks"] / self.other_data["ke"]

    def ks(self):
        return self.other_data["ks"]

    def ka(self):
        return self.other_data["ka"]


class PassiveResult(Result):
    mode = "P"

    def sel_data(self, channel=None, **kwargs):
        if channel is not None:
            kwargs.update({k: v for k, v in self.channel_map[channel].items() if k in self.data.dims})
        return self.data.sel(drop=True, **kwargs)

    def Tb(self, channel=None, **kwargs):
        """Brightness temperature; slice with e.g. frequency=37e9, polarization='V' or channel='37V'."""
        return _strongsqueeze(self.sel_data(channel=channel, **kwargs).rename("Tb"))

    def TbV(self, **kwargs):
        return _strongsqueeze(self.data.sel(polarization="V", **kwargs).rename("TbV"))

    def TbH(self, **kwargs):
        return _strongsqueeze(self.data.sel(polarization="H", **kwargs).rename("TbH"))

    def polarization_ratio(self, ratio="H_V", **kwargs):
        return _strongsqueeze(self.data.sel(polarization=ratio[0], **kwargs)
                              / self.data.sel(polarization=ratio[-1], **kwargs))

    def to_dataframe(self, channel_axis="auto", **kwargs):
        if channel_axis == "auto":
            channel_axis = "column" if self.channel_map else None
        return super().return_as_dataframe(name="Tb", channel_axis=channel_axis, **kwargs)

    def Tb_as_dataframe(self, channel_axis=None, **kwargs):
        return self.to_dataframe(channel_axis=None, **kwargs)

    def __repr__(self):
        return f"PassiveResult: TbV={self.TbV()}, TbH={self.TbH()}"


class ActiveResult(Result):
    mode = "A"

    def sel_data(self, channel=None, return_backscatter=False, **kwargs):
        if channel is not None:
            kwargs.update({k: v for k, v in self.channel_map[channel].items() if k in self.data.dims})
        if return_backscatter:
            theta = kwargs.pop("theta", None)
            theta_inc = kwargs.pop("theta_inc", None)
            if theta is not None and theta_inc is not None and not np.all(theta_inc == theta):
                raise SMRTError("theta and theta_inc must be the same when returning backscatter")
            if theta is None:
                theta = theta_inc
            if theta is None:
                theta = self.data.coords["theta_inc"]
            if np.ndim(theta) > 0:
                x = self.data.sel(drop=True, theta_inc=list(np.atleast_1d(theta)), **kwargs)
                axis = x.dims.index("theta_inc")
                shape = [1] * len(x.dims)
                shape[axis] = -1
                factor = (4 * np.pi * np.cos(np.deg2rad(np.atleast_1d(theta)))).reshape(shape)
            else:
                x = self.data.sel(drop=True, theta_inc=theta, **kwargs)
                factor = 4 * np.pi * np.cos(np.deg2rad(theta))
            x = x * factor  # sigma = 4 pi cos(theta) I (result.py:484-486)
            if return_backscatter == "dB":
                return LabeledArray(dB(x.values), list(x.coords.items()), name=x.name, attrs=x.attrs)
            return x
        return self.data.sel(drop=True, **kwargs)

    def sigma(self, channel=None, name="sigma", **kwargs):
        return _strongsqueeze(self.sel_data(channel=channel, return_backscatter="natural", **kwargs).rename(name))

    def sigma_dB(self, name="sigma_dB", channel=None, **kwargs):
        return _strongsqueeze(self.sel_data(channel=channel, return_backscatter="dB", **kwargs).rename(name))

    def to_dataframe(self, channel_axis=None, **kwargs):
        if channel_axis == "auto":
            channel_axis = "column" if self.channel_map else None
        return super().return_as_dataframe(name="sigma", channel_axis=channel_axis, return_backscatter="dB", **kwargs)

    def sigmaVV(self, name="sigmaVV", **kwargs):
        return self.sigma(polarization_inc="V", polarization="V", name=name, **kwargs)

    def sigmaHH(self, name="sigmaHH", **kwargs):
        return self.sigma(polarization_inc="H", polarization="H", name=name, **kwargs)

    def sigmaHV(self, name="sigmaHV", **kwargs):
        return self.sigma(polarization_inc="H", polarization="V", name=name, **kwargs)

    def sigmaVH(self, name="sigmaVH", **kwargs):
        return self.sigma(polarization_inc="V", polarization="H", name=name, **kwargs)

    def sigmaVV_dB(self, name="sigmaVV_dB", **kwargs):
        return dB(self.sigmaVV(name=name, **kwargs))

    def sigmaHH_dB(self, name="sigmaHH_dB", **kwargs):
        return dB(self.sigmaHH(name=name, **kwargs))

    def sigmaHV_dB(self, name="sigmaHV_dB", **kwargs):
        return dB(self.sigmaHV(name=name, **kwargs))

    def sigmaVH_dB(self, name="sigmaVH_dB", **kwargs):
        return dB(self.sigmaVH(name=name, **kwargs))

    def __repr__(self):
        return (f"ActiveResult:sigmaVV={self.sigmaVV_dB()} dB, sigmaHH={self.sigmaHH_dB()} dB, "
                f"sigmaHV={self.sigmaHV_dB()} dB")


def make_result(sensor, *args, **kwargs):
    """smrt/core/result.py:79-90."""
    if sensor.mode == "A":
        return ActiveResult(*args, channel_map=sensor.channel_map, **kwargs)
    return PassiveResult(*args, channel_map=sensor.channel_map, **kwargs)


def concat_results(result_list, coord):
    """Concatenate results along a new leading dimension (smrt/core/result.py:768-817)."""
    if not isinstance(coord, tuple):
        raise SMRTError("unknown type for the coord argument")
    dim_name, dim_value = coord
    ResultClass = type(result_list[0])
    if not all(type(r) is ResultClass for r in result_list):
        raise SMRTError("The results are not all of the same type")
    if any(r.channel_map != result_list[0].channel_map for r in result_list):
        channel_map = {ch: dict(**r.channel_map[ch], dim_name=dv) for r, dv in zip(result_list, dim_value)
                       for ch in r.channel_map}
    else:
        channel_map = result_list[0].channel_map
    data = concat_arrays([r.data for r in result_list], dim_name, dim_value)
    other = {}
    for k in result_list[0].other_data:
        arrs = [r.other_data[k] for r in result_list]
        try:
            other[k] = concat_arrays(arrs, dim_name, dim_value)
        except SMRTError:  # ragged (e.g. different layer counts): pad with NaN like xr.concat(join="outer")
            n = max(a.values.shape[0] for a in arrays_1d(arrs))
            padded = []
            for a in arrs:
                v = np.full(n, np.nan, dtype=a.values.dtype)
                v[: a.values.shape[0]] = a.values
                padded.append(LabeledArray(v, [(a.dims[0], np.arange(n))], name=a.name))
            other[k] = concat_arrays(padded, dim_name, dim_value)
    return ResultClass(data, channel_map=channel_map, other_data=other)


def arrays_1d(arrs):
    for a in arrs:
        if len(a.dims) != 1:
            raise SMRTError("only one-dimensional diagnostic arrays can be padded")
    return arrs
