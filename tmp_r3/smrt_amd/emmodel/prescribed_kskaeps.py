"""Scattering and absorption coefficients and the effective permittivity taken from the layer itself, Rayleigh phase
matrix (smrt/emmodel/prescribed_kskaeps.py): set `layer.ks`, `layer.ka` and `layer.effective_permittivity`.
Evaluated on the host like smrt_amd.emmodel.rayleigh."""
from ..core.error import SMRTError
from .rayleigh import _RayleighPhase


class Prescribed_KsKaEps(_RayleighPhase):
    def __init__(self, sensor, layer):
        missing = [a for a in ("ks", "ka", "effective_permittivity") if getattr(layer, a, None) is None]
        if missing:
            raise SMRTError(f"prescribed_kskaeps needs the layer attribute(s) {missing}")
        self.sensor, self.layer = sensor, layer
        self._ks = layer.ks
        self.ka = layer.ka
        self._effective_permittivity = layer.effective_permittivity
