"""Reflector substrate: prescribed specular reflection, emissivity 1 - R (smrt/substrate/reflector.py).  The reflection
may be a scalar or a dictionary keyed by polarisation and/or frequency; functions of the angle are not supported by the
device path.  Passive mode only, like in the reference."""
from ..core.error import SMRTError
from ..core.substrate import SubstrateBase


def make_reflector(temperature=None, specular_reflection=None):
    return Reflector(temperature=temperature, specular_reflection=specular_reflection)


class Reflector(SubstrateBase):
    device_kind = "reflector"

    def __init__(self, temperature=None, specular_reflection=None, **kwargs):
        super().__init__(temperature=temperature, **kwargs)
        self.specular_reflection = 1 if specular_reflection is None else specular_reflection

    def _get_refl(self, frequency, polarization):  # reflector.py:_get_refl
        r = self.specular_reflection
        if isinstance(r, dict):
            for key in [(frequency, polarization), (polarization, frequency), frequency, polarization]:
                if key in r:
                    r = r[key]
                    break
        if isinstance(r, dict):
            raise SMRTError("The specular_reflection argument must be a scalar or a dict with the frequency and/or "
                            "polarization as a key. If both, provide frequency and polarization as a tuple key")
        if callable(r):
            raise SMRTError("smrt_amd's Reflector takes scalar reflections (per polarisation / frequency), not functions "
                            "of the angle")
        return float(r)

    def device_params(self, frequency):
        return self._get_refl(frequency, "V"), self._get_refl(frequency, "H")
