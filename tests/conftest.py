import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


def load_golden(name):
    d = np.load(os.path.join(GOLDEN, name + ".npz"))
    return {k: d[k] for k in d.files}


def snowpack_dict(d):
    """The plain-array snowpack description stored in a fixture."""
    sp = {k: d[k] for k in ("thickness", "density", "temperature", "frac_volume")}
    sp["microstructure"] = str(d["microstructure"])
    for k in ("corr_length", "radius", "stickiness"):
        if k in d:
            sp[k] = d[k]
    return sp


def fixture_options(d):
    return dict(n_max_stream=int(d.get("opt_n_max_stream", 32)), m_max=int(d.get("opt_m_max", 2)))


PASSIVE_FIXTURES = [
    "cfg1_iba_onelayer",
    "iba_2layer_passive37",
    "cfg2_iba_L20_n32_sp0",
    "cfg2_iba_L20_n32_sp1",
    "iba_L6_n8_angles",
    "iba_L3_n16_shallow",
    "dmrt_L8_n16",
    "cfg3_dmrt_L50_n64_sp0",
]
ACTIVE_FIXTURES = ["iba_2layer_active19", "cfg4_iba_active_L5_n16", "iba_active_L4_n32_ku"]


@pytest.fixture
def golden():
    return load_golden
