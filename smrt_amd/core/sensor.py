"""Sensor configuration: the subset of smrt/core/sensor.py the DORT path consumes (passive :24-74, active :119-201,
Sensor :235-376), same constructor arguments, attributes and iteration semantics."""
import copy
from collections.abc import Sequence

import numpy as np

from .error import SMRTError, smrt_warn
from .globalconstants import C_SPEED


class SensorBase(object):
    pass


class Sensor(SensorBase):
    def __init__(self, frequency=None, theta_inc_deg=None, theta_deg=None, phi_deg=None, polarization_inc=None,
                 polarization=None, channel_map=None, name=None, wavelength=None):
        if frequency is not None:
            self.frequency = np.asarray(frequency).squeeze() if isinstance(frequency, Sequence) else frequency
            self.wavelength = C_SPEED / self.frequency
        elif wavelength is not None:
            self.wavelength = np.asarray(wavelength).squeeze() if isinstance(wavelength, Sequence) else wavelength
            self.frequency = C_SPEED / self.wavelength
        else:
            raise SMRTError("Either frequency or wavelength is required")
        self.channel_map = channel_map or dict()
        self.name = name
        self.polarization = list(polarization) if isinstance(polarization, str) else polarization
        self.polarization_inc = list(polarization_inc) if isinstance(polarization_inc, str) else polarization_inc
        if theta_deg is None:
            raise SMRTError("Sensor requires the argument 'theta_deg' to be set")
        self.theta_deg = np.atleast_1d(theta_deg).flatten().astype(dtype=float)
        if len(np.unique(self.theta_deg)) != len(self.theta_deg):
            raise SMRTError("Zenith angle theta has duplicated values which is invalid.")
        self.theta = np.radians(self.theta_deg)
        self.mu_s = np.cos(self.theta)
        if phi_deg is not None:
            self.phi_deg = np.atleast_1d(phi_deg).flatten().astype(dtype=float)
            self.phi = np.radians(self.phi_deg)
        else:
            self.phi = 0.0
        if theta_inc_deg is None:
            self.theta_inc_deg = None
            self.theta_inc = None
        else:
            self.theta_inc_deg = np.atleast_1d(theta_inc_deg).flatten().astype(dtype=float)
            if len(np.unique(self.theta_inc_deg)) != len(self.theta_inc_deg):
                raise SMRTError("Zenith angle theta_inc has duplicated values which is invalid.")
            self.theta_inc = np.radians(self.theta_inc_deg)
            self.mu_i = np.cos(self.theta_inc)

    @property
    def wavenumber(self):
        return 2 * np.pi / self.wavelength

    @property
    def mode(self):
        """"A" for active, "P" for passive (sensor.py:330-339)."""
        return "P" if self.theta_inc is None else "A"

    def basic_checks(self):
        if np.min(np.atleast_1d(self.frequency)) < 300e6:
            smrt_warn("Frequency not in microwave range: check units are Hz")

    def configurations(self):
        for axis in ["frequency", "theta_inc", "polarization_inc", "theta", "phi", "polarization"]:
            values = np.atleast_1d(getattr(self, axis))
            if len(values) > 1:
                yield axis, values

    def iterate(self, axis):
        for v in getattr(self, axis):
            sensor_subset = copy.copy(self)
            setattr(sensor_subset, axis, v)
            yield sensor_subset


def passive(frequency, theta, polarization=None, channel_map=None, name=None):
    """Generic passive microwave sensor (smrt/core/sensor.py:24-74)."""
    if polarization is None:
        polarization = ["V", "H"]
    sensor = Sensor(frequency, None, theta, None, None, polarization, channel_map=channel_map, name=name)
    sensor.basic_checks()
    return sensor


def active(frequency, theta_inc, theta=None, phi=None, polarization_inc=None, polarization=None, channel_map=None,
           name=None):
    """Generic active sensor, backscatter by default (smrt/core/sensor.py:119-201)."""
    if theta is None:
        theta = theta_inc
    if phi is None:
        phi = 180.0
    if polarization is None:
        polarization = ["V", "H"]
    if polarization_inc is None:
        polarization_inc = ["V", "H"]
    sensor = Sensor(frequency, theta_inc_deg=theta_inc, theta_deg=theta, phi_deg=phi,
                    polarization_inc=polarization_inc, polarization=polarization, channel_map=channel_map, name=name)
    sensor.basic_checks()
    return sensor
