"""Sensor catalogue entries used by the benchmark configurations (smrt/inputs/sensor_list.py:22-203,285-307)."""
import numpy as np

from ..core.error import SMRTError
from ..core.sensor import active, passive  # noqa: F401  (re-exported like the reference)


def _conical_pmw(sensor_name, frequency_dict, channel=None, frequency=None, polarization=None, theta=55, name=None):
    if frequency is None:
        frequency = sorted(set(frequency_dict.values()))
    else:
        frequency_dict = {f"{int(freq * 1e9):02d}": freq for freq in np.atleast_1d(frequency)}
    if polarization is None:
        polarization = ["H", "V"]
    channel_map = {
        freq + pola: dict(frequency=frequency_dict[freq], polarization=pola, theta=theta)
        for freq in frequency_dict for pola in polarization
    }
    if channel is not None:
        if isinstance(channel, str):
            channel = [channel]
        new_channel = []
        for ch in channel:
            new_channel += [ch + "H", ch + "V"] if ch[-1] not in "HV" else [ch]
        for ch in new_channel:
            if "18" in ch:
                channel_map[ch] = channel_map.pop("19" + ch[-1])
            if "36" in ch:
                channel_map[ch] = channel_map.pop("37" + ch[-1])
        try:
            channel_map = {ch: channel_map[ch] for ch in new_channel}
        except KeyError:
            raise SMRTError(f"{sensor_name} channel not recognized. Expected one of: {', '.join(frequency_dict)}")
    conf = {}
    for key in ("frequency", "polarization", "theta"):
        vals = []
        for cfg in channel_map.values():
            if cfg[key] not in vals:
                vals.append(cfg[key])
        conf[key] = sorted(vals) if key != "polarization" else vals
    return passive(channel_map=channel_map, name=name, **conf)


def amsre(channel=None, frequency=None, polarization=None, theta=55):
    """AMSR-E: 6.925, 10.65, 18.7, 23.8, 36.5, 89 GHz at H and V (sensor_list.py:22-64)."""
    d = {"06": 6.925e9, "10": 10.65e9, "19": 18.7e9, "23": 23.8e9, "37": 36.5e9, "89": 89e9}
    return _conical_pmw("AMSR-E", d, channel=channel, frequency=frequency, theta=theta, name="amsre")


def amsr2(channel=None, frequency=None, polarization=None, theta=55):
    """AMSR2: AMSR-E frequencies plus 7.3 GHz (sensor_list.py:67-110)."""
    d = {"06": 6.925e9, "07": 7.3e9, "10": 10.65e9, "19": 18.7e9, "23": 23.8e9, "37": 36.5e9, "89": 89e9}
    return _conical_pmw("AMSR2", d, channel=channel, frequency=frequency, theta=theta, name="asmr2")


def sentinel1(theta=None):
    """C-SAR on Sentinel 1, 5.405 GHz, 20..45 deg by 5 (sensor_list.py:285-307)."""
    if theta is None:
        theta = np.arange(20, 46, 5)
    return active(5.405e9, theta,
                  channel_map={ch: dict(polarization=ch[1], polarization_inc=ch[0]) for ch in ["HH", "VV", "HV", "VH"]},
                  name="sentinel1")
