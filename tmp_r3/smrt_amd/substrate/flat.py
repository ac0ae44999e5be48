"""Flat substrate: Fresnel reflection against a prescribed permittivity (smrt/substrate/flat.py, built by
smrt/core/interface.py:169-240 from interface/flat.py).  A cheap descriptor: the device evaluates the Fresnel terms for
every stream; the host only evaluates the permittivity model per frequency."""
from ..core.error import SMRTError
from ..core.substrate import SubstrateBase


class Flat(SubstrateBase):
    device_kind = "flat"

    def device_params(self, frequency):
        eps = self.permittivity(frequency)
        if eps is None:
            raise SMRTError("No permittivity_model have been given to the substrate 'Flat'")
        eps = complex(eps)
        return eps.real, eps.imag
