"""CPU-only tests of the host side: plugin surface, simulation flattening, Result accessors, packing, C ABI exports."""
import ctypes
import os
import re

import numpy as np
import pytest

import smrt_amd
from conftest import ROOT, load_golden
from smrt_amd import make_model, make_snowpack, sensor_list
from smrt_amd.core.error import SMRTError
from smrt_amd.core.result import ActiveResult, LabeledArray, PassiveResult, concat_results


def two_layer():
    return make_snowpack([0.1, 100], "exponential", density=[200, 400], temperature=[250.0, 250.0],
                         corr_length=[5e-5, 5e-5])


def test_make_model_resolves_plugins():
    m = make_model("iba", "dort")
    assert m.rtsolver.__name__ == "DORT" and m.emmodel.__name__ == "IBA"
    m = make_model("dmrt_qca_shortrange", "dort", rtsolver_options=dict(n_max_stream=64))
    assert m.emmodel.__name__ == "DMRT_QCA_ShortRange"
    assert "frequency" not in m.rtsolver._broadcast_capability  # DORT does not broadcast frequency (dort.py:140-146)
    with pytest.raises(SMRTError):
        make_model("iba", "no_such_solver")


def test_simulation_order_matches_reference():
    """Frequency-major, snowpack-minor flattening (model.py:485-502), pinned by a fixture from the reference."""
    d = load_golden("model_run_order")
    sps = [two_layer() for _ in range(3)]
    m = make_model("iba", "dort")
    sims, dims = m.prepare_simulations(sensor_list.amsre(), sps, None, "snowpack")
    order = [(float(se.frequency), sps.index(sp)) for se, sp in sims]
    assert [tuple(x) for x in d["order"]] == order
    assert [str(dm[0]) for dm in dims] == list(d["dims"])


def test_sensor_catalogue():
    s = sensor_list.amsre("37V")
    assert float(np.ravel(s.frequency)[0]) == 36.5e9 and s.mode == "P" and list(s.theta_deg) == [55.0]
    assert "37V" in s.channel_map
    assert len(np.atleast_1d(sensor_list.amsr2().frequency)) == 7
    s1 = sensor_list.sentinel1()
    assert s1.mode == "A" and list(s1.theta_inc_deg) == [20, 25, 30, 35, 40, 45] and np.isclose(s1.phi[0], np.pi)
    with pytest.raises(SMRTError):
        sensor_list.passive(37e9, [55, 55])


def test_snowpack_builder_scope():
    sp = make_snowpack([0.5, 0, 10], "sticky_hard_spheres", density=[250, 300, 350], temperature=265,
                       radius=[1e-4, 1e-4, 2e-4], stickiness=0.2)
    assert sp.nlayer == 2  # zero-thickness layers are dropped (make_medium.py:206-208)
    assert np.isclose(sp.layers[0].frac_volume, 250 / 916.7)
    assert sp.layers[1].microstructure.device_params == (2e-4, 0.2)
    with pytest.raises(SMRTError):
        make_snowpack([1], "exponential", density=300, corr_length=1e-4, substrate="soil")  # not a substrate object
    with pytest.raises(SMRTError):
        make_snowpack([1], "gaussian_random_field", density=300, corr_length=1e-4)
    with pytest.raises(SMRTError):
        make_snowpack([1], "exponential", density=300, corr_length=1e-4, liquid_water=0.1)


def test_substrate_atmosphere_and_emmodel_descriptors():
    """Host-side counterparts of smrt/substrate/{flat,reflector}.py, atmosphere/simple_isotropic_atmosphere.py and the
    extra emmodels: what they hand to the device batch (no GPU needed)."""
    from smrt_amd import make_atmosphere
    from smrt_amd._native import PackedBatch
    from smrt_amd.substrate.flat import Flat
    from smrt_amd.substrate.reflector import Reflector, make_reflector

    flat = Flat(temperature=270.0, permittivity_model=lambda f, t: 3.0 + 1e-10 * f * 1j)
    assert flat.device_kind == "flat" and np.allclose(flat.device_params(18.7e9), (3.0, 1.87))
    with pytest.raises(SMRTError):
        Flat(temperature=270.0).device_params(10e9)  # no permittivity model (core/interface.py:185-189)
    refl = Reflector(specular_reflection={(21e9, "H"): 0.5, "V": 0.6, 36e9: 0.7})
    assert refl.device_params(21e9) == (0.6, 0.5) and refl.device_params(36e9) == (0.7, 0.7)  # frequency key wins over polarisation (reflector.py)
    assert make_reflector(temperature=260).device_params(10e9) == (1.0, 1.0)
    with pytest.raises(SMRTError):
        Reflector(specular_reflection=np.cos).device_params(10e9)
    atm = make_atmosphere("simple_isotropic_atmosphere", tb_down={10e9: 15.0, 21e9: 23.5}, tb_up=6.0, transmittance=0.9)
    assert atm.device_params(21e9) == (23.5, 6.0, 0.9)
    with pytest.raises(SMRTError):
        atm.device_params(37e9)
    sp = atm + (two_layer() + flat)
    assert sp.substrate is flat and sp.atmosphere is atm and sp.nlayer == 2
    with pytest.raises(SMRTError):
        two_layer() + 3
    b = PackedBatch([2], [[0.1, 100]], [[0.2, 0.4]], [[250, 250]], [[5e-5, 5e-5]], None, [10e9, 21e9], np.deg2rad([55.0]),
                    substrate=("flat", [[3.0], [3.1]], [[0.1], [0.2]], [270.0]), atmosphere=([15.0, 23.5], 6.0, 0.9))
    assert b.struct.substrate_kind == 1 and b.sub_p1.shape == (2, 1) and b.atm[1].shape == (2,)
    assert bool(b.struct.atm_tb_down) and bool(b.struct.substrate_temperature)
    for name, cls in (("dmrt_qcacp_shortrange", "DMRT_QCACP_ShortRange"), ("nonscattering", "NonScattering")):
        assert make_model(name, "dort").emmodel.__name__ == cls


def test_result_accessors_passive():
    data = np.array([[[250.0, 251.0], [240.0, 241.0]], [[230.0, 231.0], [220.0, 221.0]]])  # (freq, pol, theta)
    r = PassiveResult(data, [("frequency", [19e9, 37e9]), ("polarization", ["V", "H"]), ("theta", [40.0, 55.0])],
                      channel_map={"37V": dict(frequency=37e9, polarization="V", theta=55),
                                   "19H": dict(frequency=19e9, polarization="H", theta=55)})
    assert r.TbV(frequency=37e9, theta=55) == 231.0
    assert r.TbH(frequency=19e9, theta=40) == 240.0
    assert r.Tb(channel="37V") == 231.0 and r.Tb(channel="19H") == 241.0
    assert r.TbV(theta=55).shape == (2,)
    df = r.to_dataframe(channel_axis="column", theta=55)
    assert list(df.columns) == ["37V", "19H"]
    df = r.to_dataframe(channel_axis=None)
    assert df.shape == (8, 1) and df.index.names == ["frequency", "polarization", "theta"]
    assert list(r.frequency) == [19e9, 37e9]


def test_result_accessors_active_and_concat():
    I = np.arange(18, dtype=float).reshape(3, 3, 2) * 1e-3 + 1e-3
    coords = [("polarization_inc", ["V", "H", "U"]), ("polarization", ["V", "H", "U"]), ("theta_inc", [30.0, 40.0])]
    r = ActiveResult(I.copy(), coords)
    th = np.deg2rad(40.0)
    assert np.isclose(r.sigmaVV(theta=40), 4 * np.pi * np.cos(th) * I[0, 0, 1])
    assert np.isclose(r.sigmaHV(theta=40), 4 * np.pi * np.cos(th) * I[1, 0, 1])  # labels as in result.py:625-629
    assert np.isclose(r.sigmaVV_dB(theta=30), 10 * np.log10(4 * np.pi * np.cos(np.deg2rad(30.0)) * I[0, 0, 0]))
    assert r.sigmaVV().shape == (2,)
    both = concat_results([r, ActiveResult(2 * I, coords)], ("snowpack", [0, 1]))
    assert both.data.dims == ("snowpack", "polarization_inc", "polarization", "theta_inc")
    assert np.isclose(both.sigmaVV(snowpack=1, theta=40), 2 * r.sigmaVV(theta=40))


def test_labeled_array_sel_errors():
    a = LabeledArray(np.zeros((2, 3)), [("x", [1, 2]), ("y", ["a", "b", "c"])])
    assert a.sel(x=2).dims == ("y",)
    with pytest.raises(KeyError):
        a.sel(x=5)
    with pytest.raises(KeyError):
        a.sel(z=1)


def test_packed_batch_layout():
    from smrt_amd._native import PackedBatch

    b = PackedBatch([2, 1], [[0.1, 100], [50, 1]], [[0.2, 0.4], [0.3, 0.3]], [[250, 250], [260, 260]],
                    [[5e-5, 5e-5], [1e-4, 1e-4]], None, [19e9, 37e9], np.deg2rad([55.0]))
    assert b.n_pairs == 4 and b.out_shape() == (2, 1) and b.struct.n_layers_max == 2
    assert b.thickness.flags.c_contiguous and b.thickness.dtype == np.float64
    with pytest.raises(SMRTError):
        PackedBatch([3], [[1, 1]], [[0.2, 0.2]], [[250, 250]], [[1e-4, 1e-4]], None, [19e9], [0.9])


def test_library_exports_every_declared_symbol():
    """The shared library loads on a box without GPU and exports every function include/smrt_dort.h declares."""
    from smrt_amd import _native

    header = open(os.path.join(ROOT, "include", "smrt_dort.h")).read()
    declared = set(re.findall(r"\b(smrt_[a-z0-9_]+)\s*\(", header))
    lib = ctypes.CDLL(_native.LIB_PATH)
    for sym in declared:
        assert hasattr(lib, sym), sym
    assert declared == set(_native.EXPORTED_SYMBOLS)
    assert b"gfx950" in _native.load_library().smrt_dort_version()


def test_no_cpu_fallback():
    """Without a GPU the product must fail loudly (never route through the oracle)."""
    from smrt_amd import _native

    if _native.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(SMRTError, match="no CPU fallback"):
        make_model("iba", "dort").run(sensor_list.amsre("37V"), two_layer())
    src = "".join(open(os.path.join(dp, f)).read() for dp, _, fs in os.walk(os.path.join(ROOT, "smrt_amd"))
                  for f in fs if f.endswith(".py"))
    assert "oracle" not in src.replace("never route through the oracle", "")


def test_gauss_legendre_nodes_match_scipy():
    from scipy.special import roots_legendre

    from smrt_amd._native import gauss_legendre_positive

    for n in (2, 8, 32, 64, 128):
        x, w = roots_legendre(2 * n)
        mu, wt = gauss_legendre_positive(n)
        np.testing.assert_allclose(mu, x[-1:n - 1:-1], rtol=0, atol=3e-16)
        np.testing.assert_allclose(wt, w[-1:n - 1:-1], rtol=1e-9)  # weights are not used by the path (streams.py:324-330)


def test_dort_option_validation():
    from smrt_amd.rtsolver.dort import DORT

    DORT(n_max_stream=64, diagonalization_method="half_rank_eig", error_handling="nan")
    for bad in (dict(stream_mode="uniform_air"), dict(prune_deep_snowpack=-1), dict(diagonalization_method="foo"),
                dict(error_handling="ignore"), dict(process_coherent_layers=True), dict(phase_symmetrization=True)):
        with pytest.raises(SMRTError):
            DORT(**bad)
    # prune_deep_snowpack: True is an optical depth of 6 (smrt/rtsolver/dort.py:176-178); the cache option is a no-op
    assert DORT(prune_deep_snowpack=True).prune_deep_snowpack == 6.0
    assert DORT(prune_deep_snowpack=2.5, diagonalization_cache=True).prune_deep_snowpack == 2.5
    assert DORT(prune_deep_snowpack=False).prune_deep_snowpack is None and DORT().prune_deep_snowpack is None
    from smrt_amd._native import PackedBatch

    def pb(**kw):
        return PackedBatch([1], [[1.0]], [[0.3]], [[260.0]], [[1e-4]], None, [37e9], [0.9], **kw).struct.prune_optical_depth

    assert pb() == 0.0 and pb(prune_deep_snowpack=True) == 6.0 and pb(prune_deep_snowpack=1.5) == 1.5


def test_sensor_catalogue_matches_reference():
    """amsre / amsr2 / cimr / quikscat / ascat / sentinel1 / smos / smap: same frequencies, angles, polarisations,
    channel maps (names and order), sensor names and error types as the reference's smrt/inputs/sensor_list.py
    (golden data written by tests/golden/make_sensor_golden.py from the reference)."""
    import json
    import os
    import sys

    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    sys.path.insert(0, here)
    try:
        from make_sensor_golden import describe
    finally:
        sys.path.remove(here)
    from smrt_amd.inputs import sensor_list

    with open(os.path.join(here, "sensor_catalogue.json")) as fh:
        ref = json.load(fh)
    mine = json.loads(json.dumps(describe(sensor_list), sort_keys=True))
    assert len(mine) == len(ref)
    for a, b in zip(mine, ref):
        assert a == b, (a["call"], a, b)
