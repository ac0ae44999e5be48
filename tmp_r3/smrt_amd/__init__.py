"""smrt_amd: MI355X-native DORT hot path behind SMRT's make_model() / Model.run() / Result plugin surface.

    from smrt_amd import make_model, make_snowpack, sensor_list
    m = make_model("iba", "dort")
    res = m.run(sensor_list.amsre("37V"), make_snowpack([100], "exponential", density=[320], temperature=[270],
                                                         corr_length=[5e-5]))
    print(res.TbV(), res.TbH())
"""
from .core.error import SMRTError, SMRTWarning  # noqa: F401
from .core.model import make_emmodel, make_model, make_rtsolver  # noqa: F401
from .core.plugin import register_package  # noqa: F401
from .core.result import concat_results, open_result  # noqa: F401
from .core.sensor import Sensor  # noqa: F401
from .inputs import sensor_list  # noqa: F401
from .atmosphere.simple_isotropic_atmosphere import make_atmosphere  # noqa: F401
from .inputs.make_medium import make_snow_layer, make_snowpack  # noqa: F401
from .runner.hip_batch_runner import HipBatchRunner  # noqa: F401
from .utils import dB, invdB  # noqa: F401

GHz = 1e9
cm = 1e-2
mm = 1e-3
micron = 1e-6
