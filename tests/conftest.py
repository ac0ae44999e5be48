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
ACTIVE_FIXTURES = ["iba_2layer_active19", "cfg4_iba_active_L5_n16", "iba_active_L4_n32_ku", "dmrt_active_L3_n12",
                   "iba_shs_active_L3_n8", "iba_active_L3_n10_m1_steep"]
SIGMA_RTOL = 1e-8  # backscatter, relative (BASELINE.json north_star)


def assert_backscatter_close(r, ref, rtol=SIGMA_RTOL, cross_rtol=1e-6):
    """r, ref: [..., pol, pol_inc, theta_inc].  All four V,H x V,H intensities to `rtol` relative to the co-pol level
    at that angle.  The cross-polarised terms sit 30-50 dB below co-pol and come out of a cancellation between the
    azimuth modes: on their OWN scale the reference's diagonalisation methods already differ among themselves by
    1e-8 .. 1e-7 (eig / half_rank_eig vs schur_forcedtriu, checked in tests/test_oracle_golden.py), so they are held
    to `cross_rtol` on their own scale on top of `rtol` on the co-pol scale."""
    r, ref = np.asarray(r), np.asarray(ref)
    scale = np.abs(ref[..., :2, :2, :]).max(axis=(-3, -2), keepdims=True)
    assert (np.abs(r - ref)[..., :2, :2, :] / scale).max() < rtol
    np.testing.assert_allclose(r[..., 0, 1, :], ref[..., 0, 1, :], rtol=cross_rtol, atol=0)
    np.testing.assert_allclose(r[..., 1, 0, :], ref[..., 1, 0, :], rtol=cross_rtol, atol=0)
    # third Stokes rows/columns are multiplied by sin(m pi) ~ 1e-16 in backscatter: only their level is meaningful
    assert np.abs(r[..., 2, :, :]).max() <= 10 * np.abs(ref[..., 2, :, :]).max() + 1e-30
