"""DMRT QCA short-range emmodel (smrt/emmodel/dmrt_qca_shortrange.py:53-112), host-side descriptor; see iba.py."""
from ..core.error import SMRTError
from .iba import _DeviceEMModel


class DMRT_QCA_ShortRange(_DeviceEMModel):
    device_name = "dmrt_qca_shortrange"

    def __init__(self, sensor, layer, dense_snow_correction="auto"):
        if dense_snow_correction != "auto":
            raise SMRTError("smrt_amd's DMRT_QCA_ShortRange implements dense_snow_correction='auto' only")
        if layer.microstructure_model != "sticky_hard_spheres":
            raise SMRTError("DMRT_QCA_ShortRange is only compatible with SHS microstructure model")
        super().__init__(sensor, layer)
