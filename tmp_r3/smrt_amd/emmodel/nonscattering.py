"""Non-scattering medium (smrt/emmodel/nonscattering.py): Polder-van Santen permittivity, absorption only.  Host-side
descriptor; see iba.py."""
from .iba import _DeviceEMModel


class NonScattering(_DeviceEMModel):
    device_name = "nonscattering"
