"""Isotropic atmosphere with prescribed frequency-dependent emission (up and down) and transmittance
(smrt/atmosphere/simple_isotropic_atmosphere.py).  Values are constants or dictionaries keyed by frequency."""
from ..core.error import SMRTError


class SimpleIsotropicAtmosphere:
    def __init__(self, tb_down=0.0, tb_up=0.0, transmittance=1.0):
        self.constant_tbdown = tb_down
        self.constant_tbup = tb_up
        self.constant_trans = transmittance

    @staticmethod
    def _pick(x, frequency):
        if isinstance(x, dict):
            if frequency not in x:
                raise SMRTError(f"the atmosphere has no value for the frequency {frequency}")
            x = x[frequency]
        return float(x)

    def device_params(self, frequency):
        return (self._pick(self.constant_tbdown, frequency), self._pick(self.constant_tbup, frequency),
                self._pick(self.constant_trans, frequency))

    def __add__(self, other):
        from ..core.snowpack import Snowpack

        if isinstance(other, Snowpack):
            if other.atmosphere is not None:
                raise SMRTError("stacking several atmospheres is outside the scope of smrt_amd")
            return Snowpack(layers=other.layers, interfaces=other.interfaces, substrate=other.substrate, atmosphere=self)
        raise SMRTError("Attempt to add an incorrect object to an atmopshere. Only adding an atmosphere and a snowpack "
                        "(in that order) is a valid operation.")


def make_atmosphere(atmosphere_model="simple_isotropic_atmosphere", **kwargs):
    """smrt/inputs/make_medium.py:1157-1170."""
    if atmosphere_model not in ("simple_isotropic_atmosphere", SimpleIsotropicAtmosphere):
        raise SMRTError("smrt_amd implements the 'simple_isotropic_atmosphere' only")
    return SimpleIsotropicAtmosphere(**kwargs)
