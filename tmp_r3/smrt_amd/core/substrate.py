"""Base class of the substrates (smrt/core/interface.py:86-166): a temperature and a permittivity model that is a
constant or a callable of (frequency[, temperature])."""
from .error import SMRTError


class SubstrateBase:
    device_kind = None

    def __init__(self, temperature=None, permittivity_model=None, **kwargs):
        if kwargs:
            raise SMRTError("unexpected substrate arguments: %s" % sorted(kwargs))
        self.temperature = temperature
        self.permittivity_model = permittivity_model

    def permittivity(self, frequency):
        pm = self.permittivity_model
        if pm is None:
            return None
        if callable(pm):
            try:
                return pm(frequency, self.temperature)
            except TypeError:
                return pm(frequency)
        return pm

    def __add__(self, other):  # substrate + ... is not defined; snowpack + substrate is (core/snowpack.py)
        raise SMRTError("Attempt to add an incorrect object to a substrate: use snowpack + substrate")
