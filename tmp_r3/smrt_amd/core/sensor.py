"""Sensor configurations for the DORT path.

Public surface = what the reference's plugin boundary reads from a sensor (smrt/core/sensor.py:24-74 `passive`,
:119-201 `active`, :235-376 `Sensor`, :379-420 `SensorList`): constructor arguments, the attributes `frequency`,
`wavelength`, `wavenumber`, `theta(_deg)`, `theta_inc(_deg)`, `phi(_deg)`, `mu_s`, `mu_i`, `polarization(_inc)`,
`channel_map`, `name`, `mode`, and the two methods `Model.prepare_simulations` drives: `configurations()` and
`iterate(axis)`.

Own structure: a sensor is a table of *axes* (`AXES`, in the reference's flattening order).  `axis_values(axis)` is
the one accessor, `subset(axis, index)` narrows one axis to a single value -- keeping every derived quantity
(wavelength, degrees / radians / cosines) consistent -- and `split(axes)` produces the Cartesian product of
single-valued sensors that `Model` flattens against the snowpacks (frequency-major because `AXES` starts with it)."""
import itertools

import numpy as np

from .error import SMRTError, smrt_warn
from .globalconstants import C_SPEED

AXES = ("frequency", "theta_inc", "polarization_inc", "theta", "phi", "polarization")
_ANGLE_AXES = {"theta": ("theta_deg", "mu_s"), "theta_inc": ("theta_inc_deg", "mu_i"), "phi": ("phi_deg", None)}


def _as_angle_array(deg, what):
    a = np.array(deg, dtype=float, ndmin=1).ravel()
    if np.unique(a).size != a.size:
        raise SMRTError(f"Zenith angle {what} has duplicated values which is invalid.")
    return a


def _scalar_or_array(x):
    if isinstance(x, (list, tuple)):
        return np.asarray(x).squeeze()
    return x


class SensorBase(object):
    """Marker base class: `Model.run` accepts instances of it (or sequences of them)."""


class Sensor(SensorBase):
    """One sensor configuration.  theta_inc_deg=None means passive mode.  Prefer `passive()`, `active()` or the
    catalogue in smrt_amd.inputs.sensor_list."""

    def __init__(self, frequency=None, theta_inc_deg=None, theta_deg=None, phi_deg=None, polarization_inc=None,
                 polarization=None, channel_map=None, name=None, wavelength=None):
        if frequency is None and wavelength is None:
            raise SMRTError("Either frequency or wavelength is required")
        if frequency is not None and wavelength is not None:
            smrt_warn("Sensor requires either frequency or wavelength argument, not both")
        if theta_deg is None:
            raise SMRTError("Sensor requires the argument 'theta_deg' to be set")
        self.name = name
        self.channel_map = dict(channel_map) if channel_map else {}
        self._set_frequency(_scalar_or_array(frequency) if frequency is not None
                            else C_SPEED / _scalar_or_array(wavelength))
        self.polarization = list(polarization) if isinstance(polarization, str) else polarization
        self.polarization_inc = list(polarization_inc) if isinstance(polarization_inc, str) else polarization_inc
        self._set_angle("theta", _as_angle_array(theta_deg, "theta"))
        if theta_inc_deg is None:
            self.theta_inc_deg = self.theta_inc = None
        else:
            self._set_angle("theta_inc", _as_angle_array(theta_inc_deg, "theta_inc"))
        if phi_deg is None:
            self.phi = 0.0
        else:
            self._set_angle("phi", np.array(phi_deg, dtype=float, ndmin=1).ravel())

    # ---- consistent setters ---------------------------------------------------------------------------------------
    def _set_frequency(self, f):
        self.frequency = f
        self.wavelength = C_SPEED / f

    def _set_angle(self, axis, deg):
        deg_name, mu_name = _ANGLE_AXES[axis]
        setattr(self, deg_name, deg)
        rad = np.radians(deg)
        setattr(self, axis, rad)
        if mu_name:
            setattr(self, mu_name, np.cos(rad))

    # ---- derived quantities ---------------------------------------------------------------------------------------
    @property
    def wavenumber(self):
        return 2 * np.pi / self.wavelength

    @property
    def mode(self):
        """'A' (active: an incidence angle is set) or 'P' (passive)."""
        return "A" if self.theta_inc is not None else "P"

    def basic_checks(self):
        if np.min(self.axis_values("frequency")) < 300e6:  # most likely GHz given instead of Hz
            smrt_warn("Frequency not in microwave range: check units are Hz")

    # ---- the axis table -------------------------------------------------------------------------------------------
    def axis_values(self, axis):
        """The values along one axis as a 1-d array (a scalar axis has length 1)."""
        if axis not in AXES:
            raise SMRTError(f"'{axis}' is not a sensor axis")
        return np.atleast_1d(getattr(self, axis))

    def configurations(self):
        """(axis, values) for every axis holding more than one value, in flattening order."""
        for axis in AXES:
            values = self.axis_values(axis)
            if values.size > 1:
                yield axis, values

    def subset(self, axis, index):
        """A copy narrowed to the index-th value of `axis` (a scalar for frequency and the polarisations, like the
        reference's iterate; a one-element array for the angles, whose degree / radian / cosine twins follow)."""
        values = self.axis_values(axis)
        twin = object.__new__(type(self))
        twin.__dict__.update(self.__dict__)
        if axis == "frequency":
            twin._set_frequency(values[index])
        elif axis in _ANGLE_AXES:  # narrow the degree array itself (no radians -> degrees round trip); stays an array
            twin._set_angle(axis, np.atleast_1d(getattr(self, _ANGLE_AXES[axis][0]))[index:index + 1])
        else:
            setattr(twin, axis, values[index])
        return twin

    def iterate(self, axis):
        for index in range(self.axis_values(axis).size):
            yield self.subset(axis, index)

    def split(self, axes):
        """Single-valued sensors for the Cartesian product of `axes` (first axis slowest)."""
        ranges = [range(self.axis_values(a).size) for a in axes]
        for combo in itertools.product(*ranges):
            s = self
            for a, i in zip(axes, combo):
                s = s.subset(a, i)
            yield s


class SensorList(SensorBase):
    """Several sensors run as one, stacked along `axis` ('channel': the union of their channel maps, or any sensor
    attribute such as 'frequency' or 'name') -- smrt/core/sensor.py:379-420."""

    def __init__(self, sensor_list, axis="channel"):
        self.sensor_list = list(sensor_list)
        self.axis = axis
        if axis == "channel":
            labels = self.channel
            self.channel_map = {ch: s.channel_map[ch] for s in self.sensor_list for ch in s.channel_map}
        else:
            labels = [getattr(s, axis, None) for s in self.sensor_list]
            self.channel_map = {ch: {**cfg, axis: getattr(s, axis)} for s in self.sensor_list
                                for ch, cfg in s.channel_map.items()}
        if any(lab is None for lab in labels):
            raise SMRTError(f"It is required to set '{axis}' value for each sensor")
        if len(set(labels)) != len(labels):
            raise SMRTError(f"It is required to set different '{axis}' values for each sensor")
        self._labels = np.array(labels)

    @property
    def channel(self):
        return [ch for s in self.sensor_list for ch in s.channel_map]

    @property
    def channel_list(self):
        return self.channel

    @property
    def frequency(self):
        return [s.frequency for s in self.sensor_list]

    @property
    def mode(self):
        return self.sensor_list[0].mode

    def configurations(self):
        yield self.axis, self._labels

    def iterate(self, axis=None):
        if axis not in (None, self.axis):
            raise SMRTError("SensorList is unable to iterate over a different axis than its axis")
        return iter(self.sensor_list)


def passive(frequency, theta, polarization=None, channel_map=None, name=None):
    """Generic passive microwave sensor: frequency (Hz), viewing angle(s) from nadir (degrees), polarizations
    (default V and H)."""
    sensor = Sensor(frequency=frequency, theta_deg=theta, polarization=polarization or ["V", "H"],
                    channel_map=channel_map, name=name)
    sensor.basic_checks()
    return sensor


def active(frequency, theta_inc, theta=None, phi=None, polarization_inc=None, polarization=None, channel_map=None,
           name=None):
    """Generic active sensor; with theta / phi unset it is the backscatter configuration (theta = theta_inc,
    phi = 180 degrees)."""
    sensor = Sensor(frequency=frequency, theta_inc_deg=theta_inc, theta_deg=theta_inc if theta is None else theta,
                    phi_deg=180.0 if phi is None else phi, polarization_inc=polarization_inc or ["V", "H"],
                    polarization=polarization or ["V", "H"], channel_map=channel_map, name=name)
    sensor.basic_checks()
    return sensor
