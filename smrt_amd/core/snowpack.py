"""Snowpack container (counterpart of smrt/core/snowpack.py:34-260 for Flat interfaces without substrate)."""
import numpy as np

from ..interface.flat import Flat
from .error import SMRTError


class Snowpack:
    def __init__(self, layers=None, interfaces=None, substrate=None, atmosphere=None):
        if substrate is not None:
            raise SMRTError("substrates are outside the scope of smrt_amd (semi-infinite bottom layer, SURVEY 8f)")
        if atmosphere is not None:
            raise SMRTError("atmospheres are outside the scope of smrt_amd (SURVEY 8f)")
        self.layers = list(layers) if layers is not None else []
        self.interfaces = list(interfaces) if interfaces is not None else [Flat() for _ in self.layers]
        self.substrate = None
        self.atmosphere = None

    @property
    def nlayer(self):
        return len(self.layers)

    @property
    def layer_thicknesses(self):
        return np.array([lay.thickness for lay in self.layers])

    @property
    def layer_densities(self):
        return np.array([lay.density for lay in self.layers])

    def profile(self, property_name):
        return np.array([getattr(lay, property_name) for lay in self.layers])

    def append(self, layer, interface=None):
        if interface is not None and not isinstance(interface, Flat):
            raise SMRTError("only Flat interfaces are in the scope of smrt_amd")
        self.layers.append(layer)
        self.interfaces.append(interface or Flat())
