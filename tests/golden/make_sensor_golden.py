#!/usr/bin/env python
"""Golden vector of the instrument catalogue: what the reference's smrt/inputs/sensor_list.py returns for a list of
calls (frequencies, angles, polarisations, channel maps, names, error types), stored as data in sensor_catalogue.json.
RUNS ONLY IN THE BUILD CONTAINER (needs /root/reference); tests/test_host_logic.py runs `describe` on
smrt_amd.inputs.sensor_list and compares.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_sensor_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

CALLS = [("amsre", (), {}), ("amsre", ("37V",), {}), ("amsre", ("36",), {}), ("amsre", (["18H", "89"],), {}),
         ("amsre", (), dict(theta=50)), ("amsre", (), dict(frequency=[10e9, 37e9])), ("amsre", ("99V",), {}),
         ("amsr2", (), {}), ("amsr2", ("07H",), {}), ("cimr", (), {}), ("cimr", ("01V",), {}),
         ("quikscat", (), {}), ("quikscat", ("HH46",), {}), ("quikscat", (), dict(theta=54)),
         ("ascat", (), {}), ("ascat", ([30, 40],), {}), ("sentinel1", (), {}), ("sentinel1", ([30],), {}),
         ("smos", (), {}), ("smos", ([10, 20],), {}), ("smap", ("P",), {}), ("smap", ("A",), {}), ("smap", ("A", 35), {}),
         ("smap", ("X",), {})]


def _norm(x):
    if x is None:
        return None
    a = np.atleast_1d(x)
    return [str(v) if a.dtype.kind in "US" else float(v) for v in a]


def describe(module):
    out = []
    for name, args, kw in CALLS:
        rec = dict(call=[name, repr(args), repr(sorted(kw.items()))])
        try:
            s = getattr(module, name)(*args, **kw)
        except Exception as e:  # noqa: BLE001  (the error TYPE is part of the behaviour)
            rec["error"] = type(e).__name__
            out.append(rec)
            continue
        cm = s.channel_map
        rec.update(frequency=_norm(s.frequency), theta=_norm(s.theta_deg), theta_inc=_norm(s.theta_inc_deg),
                   polarization=_norm(s.polarization), polarization_inc=_norm(s.polarization_inc), mode=s.mode,
                   name=s.name, channels=None if cm is None else list(cm),
                   channel_map=None if cm is None else {k: {kk: _norm(vv) for kk, vv in sorted(v.items())}
                                                        for k, v in cm.items()})
        out.append(rec)
    return out


if __name__ == "__main__":
    sys.dont_write_bytecode = True
    sys.path.insert(0, os.path.join(HERE, "_refstubs"))
    sys.path.insert(0, "/root/reference")
    from smrt.inputs import sensor_list

    with open(os.path.join(HERE, "sensor_catalogue.json"), "w") as fh:
        json.dump(describe(sensor_list), fh, indent=1, sort_keys=True)
    print("wrote sensor_catalogue.json (%d calls)" % len(CALLS))
