"""Instrument catalogue: the named sensors of the reference (smrt/inputs/sensor_list.py) as a table.

Every conical-scanning radiometer is one row of `_RADIOMETERS` (channel prefix -> frequency); the channel grammar of
the reference (two-digit frequency prefix, optional H|V suffix, 18 == 19 GHz, 36 == 37 GHz, sensor_list.py:149-203)
is resolved by `_resolve_channels`.  Radars and L-band radiometers are small functions below.  The values are
instrument facts: AMSR-E / AMSR2 / CIMR channel frequencies, QuikSCAT 13.4 GHz at 46 (HH) and 54 (VV) degrees, ASCAT
5.255 GHz VV, Sentinel-1 5.405 GHz, SMOS 1.41 GHz, SMAP 1.4 GHz radiometer / 1.26 GHz radar.
"""
import numpy as np

from ..core.error import SMRTError
from ..core.sensor import active, passive  # noqa: F401  (re-exported like the reference)

# name -> (display name, default incidence angle, {channel prefix: frequency in Hz})
_RADIOMETERS = {
    "amsre": ("AMSR-E", 55, {"06": 6.925e9, "10": 10.65e9, "19": 18.7e9, "23": 23.8e9, "37": 36.5e9, "89": 89e9}),
    "amsr2": ("AMSR2", 55, {"06": 6.925e9, "07": 7.3e9, "10": 10.65e9, "19": 18.7e9, "23": 23.8e9, "37": 36.5e9,
                             "89": 89e9}),
    "cimr": ("CIMR", 55, {"01": 1.4135e9, "06": 6.925e9, "10": 10.65e9, "19": 18.7e9, "37": 36.5e9}),
}
_ALIASES = {"18": "19", "36": "37"}  # interchangeable channel prefixes of the reference


def _resolve_channels(display, table, wanted, polarizations):
    """Channel names asked for -> {name as asked: (frequency, polarization)}; a name without suffix means H and V."""
    if isinstance(wanted, str):
        wanted = [wanted]
    out = {}
    for ch in wanted:
        body, pols = (ch[:-1], [ch[-1]]) if ch[-1] in "HV" else (ch, ["H", "V"])
        key = _ALIASES.get(body, body)
        if key not in table or any(p not in polarizations for p in pols):
            raise SMRTError(f"{display} channel not recognized. Expected one of: {', '.join(table)}")
        for p in pols:
            out[body + p] = (table[key], p)
    return out


def common_conical_pmw(sensor_name, frequency_dict, channel=None, frequency=None, polarization=None, theta=55, name=None):
    """A conical-scanning radiometer from its channel table (role of sensor_list.py:149-203): all the channels, the
    listed `channel` names, or -- with `frequency` -- ad-hoc channels named by their integer GHz value."""
    table = dict(frequency_dict)
    if frequency is not None:
        table = {"%02d" % int(f * 1e9): f for f in np.atleast_1d(frequency)}  # (sic: the reference's naming rule)
    pols = ["H", "V"] if polarization is None else list(polarization)
    if channel is None:
        picked = {k + p: (f, p) for k, f in table.items() for p in pols}
    else:
        picked = _resolve_channels(sensor_name, table, channel, pols)
    channel_map = {ch: dict(frequency=f, polarization=p, theta=theta) for ch, (f, p) in picked.items()}
    freqs = sorted({f for f, _ in picked.values()})
    seen_pols = [p for p in ("H", "V") if any(q == p for _, q in picked.values())]
    return passive(freqs, [theta] if np.ndim(theta) == 0 else sorted(set(theta)), seen_pols, channel_map=channel_map,
                   name=name)


def _radiometer(key, channel, frequency, polarization, theta, name):
    display, default_theta, table = _RADIOMETERS[key]
    return common_conical_pmw(display, table, channel=channel, frequency=frequency, polarization=polarization,
                              theta=default_theta if theta is None else theta, name=name)


def amsre(channel=None, frequency=None, polarization=None, theta=55):
    """AMSR-E: 6.925, 10.65, 18.7, 23.8, 36.5, 89 GHz at H and V; channels "06H" .. "89V" (sensor_list.py:22-64)."""
    return _radiometer("amsre", channel, frequency, None, theta, "amsre")


def amsr2(channel=None, frequency=None, polarization=None, theta=55):
    """AMSR2: the AMSR-E frequencies plus 7.3 GHz (sensor_list.py:67-110; the reference names it "asmr2")."""
    return _radiometer("amsr2", channel, frequency, None, theta, "asmr2")


def cimr(channel=None, frequency=None, polarization=None, theta=55):
    """CIMR: 1.4135, 6.925, 10.65, 18.7, 36.5 GHz at H and V (sensor_list.py:113-146)."""
    return _radiometer("cimr", channel, frequency, None, theta, "cimr")


def filter_channel_map(channel_map, channel):
    """The entries of `channel_map` named by `channel` (a name or a list; sensor_list.py:378-383)."""
    names = [channel] if isinstance(channel, str) else list(channel)
    return {ch: channel_map[ch] for ch in names}


def extract_configuration(channel_map):
    """Distinct values of every sensor argument over the channels, scalars when unique (sensor_list.py:386-399)."""
    conf = {}
    for key in ("frequency", "polarization", "theta", "polarization_inc", "theta_inc"):
        vals = [cfg[key] for cfg in channel_map.values() if key in cfg]
        if len(vals) != len(channel_map) or not vals:
            continue
        u = np.unique(vals)
        conf[key] = u[0] if len(u) == 1 else u
    return conf


def quikscat(channel=None, theta=None):
    """SeaWinds on QuikSCAT, 13.4 GHz: HH at 46 deg and VV at 54 deg (channels "HH46", "VV54"; sensor_list.py:206-251)."""
    beams = {"HH46": 46, "VV54": 54}
    if channel is None:
        angles = [46, 54] if theta is None else list(np.atleast_1d(theta))
        channel = [ch for ch, a in beams.items() if a in angles]
    cmap = filter_channel_map({ch: dict(polarization=ch[1], polarization_inc=ch[0], theta=a, theta_inc=a)
                               for ch, a in beams.items()}, channel)
    if theta is None:
        theta = sorted({cfg["theta"] for cfg in cmap.values()})
    return active(13.4e9, theta, polarization_inc=["V", "H"], polarization=["V", "H"], channel_map=cmap, name="quikscat")


def ascat(theta=None):
    """ASCAT, 5.255 GHz, VV, 25..65 deg by 5 unless given (sensor_list.py:254-282)."""
    if theta is None:
        theta = np.arange(25, 70, 5)
    cmap = {f"VV{t}": dict(polarization_inc="V", polarization="V", theta=t, theta_inc=t) for t in np.atleast_1d(theta)}
    return active(5.255e9, theta, polarization_inc="V", polarization="V", channel_map=cmap, name="ascat")


def sentinel1(theta=None):
    """C-SAR on Sentinel 1, 5.405 GHz, 20..45 deg by 5 (sensor_list.py:285-307)."""
    if theta is None:
        theta = np.arange(20, 46, 5)
    cmap = {ch: dict(polarization=ch[1], polarization_inc=ch[0]) for ch in ("HH", "VV", "HV", "VH")}  # co-pol first
    return active(5.405e9, theta, channel_map=cmap, name="sentinel1")


def smos(theta=None):
    """MIRAS on SMOS, 1.41 GHz, 0..60 deg by 5 unless given (sensor_list.py:310-330)."""
    if theta is None:
        theta = np.arange(0, 61, 5)
    return passive(1.41e9, theta, name="smos",
                   channel_map={"01" + p: dict(polarization=p, theta=55) for p in "HV"})


def smap(mode, theta=40):
    """SMAP: the 1.4 GHz radiometer (mode "P") or the 1.26 GHz radar (mode "A") at 40 deg (sensor_list.py:333-368)."""
    if mode == "P":
        return passive(1.4e9, theta=theta, channel_map={"01" + p: dict(polarization=p) for p in "HV"}, name="smap")
    if mode == "A":
        return active(1.26e9, theta, theta=theta, name="smap",
                      channel_map={ch: dict(polarization=ch[1], polarization_inc=ch[0]) for ch in ("HH", "VV", "HV")})
    raise SMRTError("mode must be 'A' or 'P'")


def cristal_amrcr(channel):
    """Not defined by the reference either (sensor_list.py:370-375)."""
    raise NotImplementedError()
