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


def test_simulation_plan_shapes_and_zip_mode():
    """The flattened grid as index vectors (smrt_amd/core/model.py:SimulationPlan): frequency-major for one sensor x
    many snowpacks, pairwise for a sequence of sensors, no snowpack dimension for a single snowpack; dict / Series /
    DataFrame inputs name the snowpack dimension like the reference (model.py:415-470)."""
    import pandas as pd

    from smrt_amd.core.model import SimulationPlan

    m = make_model("iba", "dort")
    sps = [two_layer() for _ in range(4)]
    plan = m.plan(sensor_list.amsre(), sps)
    assert isinstance(plan, SimulationPlan) and len(plan) == 6 * 4 and plan.shape == (6, 4)
    assert [d[0] for d in plan.dimensions] == ["frequency", "snowpack"]
    assert list(plan.sensor_index) == sorted(plan.sensor_index) and list(plan.snowpack_index[:4]) == [0, 1, 2, 3]
    assert all(np.ndim(s.frequency) == 0 for s in plan.sensors)
    one = m.plan(sensor_list.amsre("37V"), sps[0])
    assert len(one) == 1 and one.dimensions == [] and one.scalar_snowpack
    zipped = m.plan([sensor_list.passive(f, 55) for f in (10e9, 19e9, 37e9, 89e9)], sps)
    assert len(zipped) == 4 and list(zipped.sensor_index) == list(zipped.snowpack_index) == [0, 1, 2, 3]
    with pytest.raises(SMRTError):
        m.plan([sensor_list.passive(37e9, 55)] * 3, sps)        # lengths differ
    with pytest.raises(SMRTError):
        m.plan([sensor_list.amsre()] * 4, sps)                  # zip mode needs single-configuration sensors
    named = m.plan(sensor_list.amsre("37V"), {"a": sps[0], "b": sps[1]})
    assert named.dimensions[0][0] == "snowpack" and list(named.dimensions[0][1]) == ["a", "b"]
    ser = pd.Series(sps[:2], index=pd.Index([2020, 2021], name="year"))
    assert m.plan(sensor_list.amsre("37V"), ser).dimensions[0] == ("year", [2020, 2021])
    df = pd.DataFrame({"snowpack": sps[:2], "site": ["x", "y"]})
    assert len(m.plan(sensor_list.amsre("37V"), df)) == 2
    with pytest.raises(SMRTError):
        m.plan(sensor_list.amsre("37V"), df, snowpack_column="nope")
    with pytest.raises(SMRTError):
        m.plan(sensor_list.amsre("37V"), sps, snowpack_dimension=("depth", [1, 2]))  # wrong number of labels


def test_sensor_axes_subset_and_sensorlist():
    """Sensor as a table of axes: subsets keep the derived quantities consistent; SensorList stacks single-configuration
    sensors along 'channel' or an attribute (smrt/core/sensor.py:379-420)."""
    from smrt_amd.core.sensor import Sensor, SensorList

    s = sensor_list.passive([10e9, 37e9], [30, 55])
    assert [a for a, _ in s.configurations()] == ["frequency", "theta", "polarization"]
    sub = list(s.iterate("frequency"))
    assert [float(x.frequency) for x in sub] == [10e9, 37e9]
    assert np.isclose(sub[1].wavelength, 299792458.0 / 37e9) and np.isclose(sub[1].wavenumber, 2 * np.pi * 37e9 / 299792458.0)
    t = s.subset("theta", 1)
    assert list(t.theta_deg) == [55.0] and np.isclose(t.mu_s[0], np.cos(np.deg2rad(55))) and list(s.theta_deg) == [30.0, 55.0]
    assert len(list(s.split(["frequency", "theta"]))) == 4
    with pytest.raises(SMRTError):
        s.axis_values("altitude")
    with pytest.raises(SMRTError):
        Sensor(theta_deg=55)                       # neither frequency nor wavelength
    assert np.isclose(Sensor(wavelength=0.21, theta_deg=40).frequency, 299792458.0 / 0.21)
    a = sensor_list.active(13e9, [30, 40])
    assert a.mode == "A" and np.allclose(a.phi, np.pi) and list(a.theta_deg) == list(a.theta_inc_deg)
    sl = SensorList([sensor_list.amsre("19V"), sensor_list.amsre("37H")])
    assert sl.channel == ["19V", "37H"] and list(dict(sl.configurations())["channel"]) == ["19V", "37H"]
    assert len(list(sl.iterate())) == 2 and sl.mode == "P"
    with pytest.raises(SMRTError):
        SensorList([sensor_list.amsre("19V"), sensor_list.amsre("19V")])
    with pytest.raises(SMRTError):
        list(sl.iterate("frequency"))
    byname = SensorList([sensor_list.passive(19e9, 55, name="a"), sensor_list.passive(37e9, 55, name="b")], axis="name")
    assert list(dict(byname.configurations())["name"]) == ["a", "b"]
    plan = make_model("iba", "dort").plan(sl, [two_layer(), two_layer()])
    assert plan.shape == (2, 2) and plan.dimensions[0][0] == "channel"


def test_emmodel_configuration_is_honoured_on_the_batch_path():
    """Per-layer emmodels and emmodel options reach the same checks on the batching runner as on the per-simulation
    route (Model.prepare_emmodels): nothing is silently dropped (hip_batch_runner ignored them)."""
    from smrt_amd.core.layer import Layer
    from smrt_amd.core.snowpack import Snowpack
    from smrt_amd.rtsolver.dort import DORT

    sp = two_layer()
    m = make_model("iba", "dort", emmodel_options=dict(dense_snow_correction="auto"))
    plan = m.plan(sensor_list.amsre("37V"), [sp])
    assert DORT.emmodel_names(m, plan) == "iba"           # "auto" is a no-op below frac_volume 0.5 (smrt/emmodel/iba.py:85-105)
    dense = Snowpack(layers=[Layer(0.1, "exponential", 200, 250.0, corr_length=5e-5),
                             Layer(10.0, "exponential", 600, 250.0, corr_length=5e-5)])   # frac_volume 0.65: needs the inversion
    plan_d = m.plan(sensor_list.amsre("37V"), [dense, sp])
    # above half ice the layer goes to the device's inverted medium (SMRT_EM_IBA_INVERTED), layer by layer
    assert DORT.emmodel_names(m, plan_d) == [["iba", "iba_inverted"], ["iba", "iba"]]
    assert DORT.emmodel_names(make_model("iba", "dort"), plan_d) == "iba"   # without the option: plain IBA, as in the reference
    with pytest.raises(SMRTError, match="dense_snow_correction"):
        make_model("iba", "dort", emmodel_options=dict(dense_snow_correction="yes")).prepare_emmodels(plan.sensors[0], sp)
    mixed = Snowpack(layers=[Layer(0.1, "exponential", 200, 250.0, corr_length=5e-5),
                             Layer(10.0, "exponential", 400, 250.0, corr_length=5e-5, emmodel="nonscattering")])
    m2 = make_model("iba", "dort")                            # a layer with its own emmodel: a heterogeneous snowpack
    assert DORT.emmodel_names(m2, m2.plan(sensor_list.amsre("37V"), [mixed, sp])) == [["iba", "nonscattering"], ["iba", "iba"]]
    m3 = make_model(["iba", "iba"], "dort")                   # a per-layer list of one kind is the uniform case
    assert DORT.emmodel_names(m3, m3.plan(sensor_list.amsre("37V"), [sp])) == "iba"
    m4 = make_model({"snow": "iba", "ice": "nonscattering"}, "dort")   # per medium
    icy = Snowpack(layers=[Layer(0.1, "exponential", 200, 250.0, corr_length=5e-5),
                           Layer(10.0, "exponential", 900, 250.0, corr_length=5e-5, medium="ice")])
    assert DORT.emmodel_names(m4, m4.plan(sensor_list.amsre("37V"), [icy])) == [["iba", "nonscattering"]]
    class Foreign:   # a class without a device implementation is evaluated on the host: (class, options) per layer
        def __init__(self, sensor, layer):
            pass
    mf = make_model(Foreign, "dort")
    assert DORT.emmodel_names(mf, mf.plan(sensor_list.amsre("37V"), [sp])) == [[(Foreign, {}), (Foreign, {})]]
    with pytest.raises(SMRTError):
        Snowpack(layers=sp.layers, interfaces=["rough", "rough"])   # interfaces are validated by the constructor too (protocol objects pass)


def test_ctypes_struct_matches_the_library():
    """smrt_dort_abi reports sizeof(smrt_batch) and every field offset as compiled; the ctypes declaration and the
    stub printed in INTEGRATION.md must agree with it (the documented stub had gone stale)."""
    from smrt_amd import _native

    lib = _native.load_library()          # load_library itself refuses a mismatching layout
    theirs = _native.abi_layout(lib)
    mine = [ctypes.sizeof(_native.SmrtBatch)] + [getattr(_native.SmrtBatch, n).offset for n, _ in _native.SmrtBatch._fields_]
    assert mine == theirs and len(theirs) == len(_native.SmrtBatch._fields_) + 1
    # field NAMES and ORDER of the ctypes declaration == the header's struct
    header = open(os.path.join(ROOT, "include", "smrt_dort.h")).read()
    body = header[header.index("typedef struct smrt_batch {"):header.index("} smrt_batch;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = re.findall(r"(\w+)\s*;", body)
    assert names == [n for n, _ in _native.SmrtBatch._fields_]
    # the stub a maintainer would copy from INTEGRATION.md is generated from the header: it must be current, and it
    # must run as printed against the library (its last line is the layout assertion)
    import importlib.util

    spec = importlib.util.spec_from_file_location("gen_ctypes_stub", os.path.join(ROOT, "tools", "gen_ctypes_stub.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    stub = gen.stub()
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    block = doc[doc.index(gen.BEGIN) + len(gen.BEGIN):doc.index(gen.END)]
    assert block.strip() == ("```python\n" + stub + "\n```").strip(), "run `python tools/gen_ctypes_stub.py --update`"
    exec(compile(stub.replace('"libsmrt_dort.so"', repr(_native.LIB_PATH)), "<INTEGRATION.md stub>", "exec"), {})


def test_shard_by_cost_balances_work():
    from smrt_amd.rtsolver.dort import shard_by_cost

    cost = np.r_[np.full(100, 8.0), np.full(100, 1.0)]          # first half eight times as expensive
    b = shard_by_cost(cost, 4)
    assert b[0] == 0 and b[-1] == 200 and np.all(np.diff(b) > 0)
    per = [cost[b[k]:b[k + 1]].sum() for k in range(4)]
    assert max(per) - min(per) <= 2 * 8.0                       # every cut within one item of the ideal one
    assert list(shard_by_cost(np.ones(10), 1)) == [0, 10]


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
    wet = make_snowpack([1], "exponential", density=300, corr_length=1e-4, temperature=273.15, liquid_water=0.1)   # wet snow: round 4
    assert wet.layers[0].liquid_water == 0.1 and np.isclose(wet.layers[0].frac_volume, 300 / (916.7 * 0.9 + 1000 * 0.1))
    with pytest.raises(SMRTError):
        make_snowpack([1], "exponential", density=300, corr_length=1e-4, salinity=0.01)


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


def test_result_save_and_open(tmp_path):
    """Result.save / open_result (smrt/core/result.py:52-76,138-147): a netCDF file laid out like xarray's
    DataArray.to_netcdf -- one data variable, coordinate variables, character arrays for the polarisations."""
    from scipy.io import netcdf_file

    from smrt_amd import open_result

    vals = np.arange(2 * 3 * 2 * 2, dtype=float).reshape(2, 3, 2, 2) + 200.25
    res = PassiveResult(vals, [("frequency", [19e9, 37e9]), ("snowpack", [0, 1, 2]), ("polarization", ["V", "H"]),
                               ("theta", [40.0, 55.0])])
    path = str(tmp_path / "res.nc")
    res.save(path)
    back = open_result(path)
    assert isinstance(back, PassiveResult) and back.data.dims == res.data.dims
    assert np.array_equal(back.data.values, vals) and list(back.data.coords["polarization"]) == ["V", "H"]
    assert float(back.TbV(frequency=37e9, snowpack=2, theta=55)) == vals[1, 2, 0, 1]
    with netcdf_file(path, "r", mmap=False) as nc:
        assert "__xarray_dataarray_variable__" in nc.variables and nc.variables["polarization"][:].dtype.kind == "S"
    act = ActiveResult(np.ones((3, 3, 2)), [("polarization_inc", ["V", "H", "U"]), ("polarization", ["V", "H", "U"]),
                                            ("theta_inc", [30.0, 40.0])])
    act.save(str(tmp_path / "act.nc"))
    back = open_result(str(tmp_path / "act.nc"))
    assert isinstance(back, ActiveResult) and np.isclose(back.sigmaVV(theta_inc=40.0), 4 * np.pi * np.cos(np.deg2rad(40.0)))


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
                dict(error_handling="ignore")):
        with pytest.raises(SMRTError):
            DORT(**bad)
    assert DORT(process_coherent_layers=True).process_coherent_layers is True
    assert DORT(phase_symmetrization=True).phase_symmetrization is True   # a no-op for the device emmodels (see DORT)
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


def test_kernel_occupancy_as_designed():
    """The pipeline kernels are designed for a number of resident workgroups per CU: two for the prep and
    finish kernels, four or more for the Jacobi kernel (its size classes: as many as their LDS lets in).  What decides it besides LDS is the register count the COMPILER ends up
    with -- a helper function that spills into accumulation registers silently halves it (it happened: 288 registers,
    one workgroup per CU, 50 ms instead of 29 ms).  The build keeps the compiler's resource remarks per translation unit
    (smrt_amd/csrc/build/*.resources.txt, -Rpass-analysis=kernel-resource-usage); this test reads them."""
    import glob

    files = sorted(glob.glob(os.path.join(ROOT, "smrt_amd", "csrc", "build", "*.resources.txt")))
    if not files:
        pytest.skip("library not built here (prebuilt .so only)")
    waves = {}
    for f in files:
        for blk in open(f).read().split(" Function Name: ")[1:]:
            name = blk.split()[0]
            waves[name] = int(re.search(r"Occupancy \[waves/SIMD\]: (\d+)", blk).group(1))
    want = {"dort_finish2_kernelILi256E": 2, "dort_prep_kernelILi256E": 3, "dort_jacobi_kernelILi256ELi0ELi128E": 4,
            # the size classes of the Jacobi kernel (k_jacobi.hip): 7 / 6 / 4 workgroups of 3 / 4 / 4 wavefronts per CU by LDS
            "dort_jacobi_kernelILi128ELi0ELi32E": 7, "dort_jacobi_kernelILi192ELi32ELi48E": 6,
            "dort_jacobi_kernelILi256ELi48ELi56E": 6, "dort_jacobi_kernelILi256ELi56ELi64E": 4,
            "dort_active_finish_kernelILi256E": 2, "dort_active_prep_kernelILi256E": 2, "dort_finish_kernel_gmemILi256E": 2,
            "dort_active_finish_kernel_gmemILi256E": 2, "dort_passive_big_kernelILi256ELi6ELi2E": 2,
            "dort_active_big_kernelILi256ELi6ELi2E": 2, "dort_jacobi_big_kernel": 3,
            "dort_finish_reg_kernel": 1,   # one wavefront per SIMD by design: the whole register file
            # the strip finish kernels: eight wavefronts, one workgroup per CU / four wavefronts, three workgroups per CU
            "dort_finish_strip_kernel": 2, "dort_finish_strip4_kernel": 3, "dort_finish_strip4_direct_kernel": 3, "dort_prep_kernel_wide": 2,
            # the symmetric eigensolver (k_eig.hip), one wavefront per item and no scratch: the register rows of its largest
            # size class (64 rows: 128 registers for the row alone) still leave two wavefronts per SIMD
            "dort_eig_tridiag_kernel": 2, "dort_eig_vectors_kernel": 2, "dort_eig_gram_kernel": 2, "dort_eig_chase_kernel": 4}
    for key, minimum in want.items():
        hits = [v for k, v in waves.items() if key in k]
        assert hits, key
        assert min(hits) >= minimum, (key, hits)
    scratch = {}
    for f in files:
        for blk in open(f).read().split(" Function Name: ")[1:]:
            scratch[blk.split()[0]] = int(re.search(r"ScratchSize \[bytes/lane\]: (\d+)", blk).group(1))
    assert all(v == 0 for k, v in scratch.items() if "dort_eig_" in k), {k: v for k, v in scratch.items() if "dort_eig_" in k}
    # the register-resident finish kernel must fit four wavefronts in the LDS of a CU at the headline shape (32 streams,
    # 20 layers): matrix slot + exchange vectors + tables <= 40 KB
    from smrt_amd import _native
    lib = _native.load_library()
    assert lib.smrt_dort_finish_reg_lds_bytes(32, 20) <= 40 * 1024
    # the strip finish kernels: three four-wavefront workgroups per CU at the headline shape, the eight-wavefront one within
    # the 160 KB of a CU at the configs[2] shape (64 streams, 50 layers)
    assert 3 * lib.smrt_dort_finish_strip_lds_bytes(32, 20, 4) <= 160 * 1024
    assert lib.smrt_dort_finish_strip_lds_bytes(64, 50, 8) <= 160 * 1024 and lib.smrt_dort_finish_strip_lds_bytes(64, 50, 3) == -1
    # the size classes of the Jacobi kernel at the headline shape: 7 / 6 / 4 workgroups in the 160 KB of a CU, each with
    # the layout of its own largest item (one layout for all would hold 4), bank-conflict-free leading dimensions
    for columns, per_cu in ((32, 14), (48, 7), (56, 6), (64, 4), (0, 4)):
        assert lib.smrt_dort_jacobi_lds_bytes(32, 2, columns) * per_cu <= 160 * 1024, columns
    assert lib.smrt_dort_jacobi_lds_bytes(32, 2, 64) == lib.smrt_dort_jacobi_lds_bytes(32, 2, 0)
    assert lib.smrt_dort_jacobi_lds_bytes(64, 2, 0) <= 160 * 1024 < 2 * lib.smrt_dort_jacobi_lds_bytes(64, 2, 0)
    assert lib.smrt_dort_jacobi_lds_bytes(64, 3, 0) == -1


def test_host_evaluated_emmodels_are_packed_for_the_device():
    """DORT._evaluate_on_host (no GPU needed): shapes and contents of the SMRT_EM_HOST arrays, the compressed ordering of
    the phase-matrix modes, and the refusals (anisotropic ks, a phase matrix without reciprocity)."""
    from conftest import host_batch_from_fixture, load_golden
    from oracle import dort_oracle as O
    from smrt_amd import make_snowpack
    from smrt_amd.core.error import SMRTError
    from smrt_amd.core.sensor import passive
    from smrt_amd.emmodel.rayleigh import Rayleigh
    from smrt_amd.rtsolver.dort import DORT

    d = load_golden("rayleigh_L3_n12_active")
    b = host_batch_from_fixture(d)
    L, n, m_max = 3, 12, 2
    assert b.host_layer.shape == (1, L, 4) and b.host_streams.shape == (1, L) and b.host_phase.shape == (1, L, m_max + 1, 2, 3 * n, 3 * n)
    assert int(b.struct.emmodel) == 4 and (b.layer_kind == 4).all()
    np.testing.assert_allclose(b.host_layer[0, :, 0], d["f0_ks"], rtol=1e-12)
    np.testing.assert_allclose(b.host_layer[0, :, 1], d["f0_ka"], rtol=1e-12)
    # against the oracle's statement of the reference's table (rayleigh.py:52-127), layer 1, every mode
    sp = dict(thickness=d["thickness"], density=d["density"], temperature=d["temperature"], frac_volume=d["frac_volume"],
              microstructure="independent_sphere", radius=d["radius"])
    ems = O.make_layers("rayleigh", float(d["frequency"][0]), sp)
    st = O.compute_streams(n, np.array([e.eps_eff for e in ems]))
    nl = int(b.host_streams[0, 1])
    assert nl == st.n[1]
    full = np.concatenate((st.mu[1], -st.mu[1]))
    ft = ems[1].ft_even_phase(full, full, m_max, 3)
    for m in range(m_max + 1):
        Cm = O.compress(ft[:, :, m])
        np.testing.assert_allclose(b.host_phase[0, 1, m, 0, :3 * nl, :3 * nl], Cm[:3 * nl, :3 * nl], rtol=1e-12, atol=1e-18)
        np.testing.assert_allclose(b.host_phase[0, 1, m, 1, :3 * nl, :3 * nl], Cm[:3 * nl, 3 * nl:], rtol=1e-12, atol=1e-18)

    sp2 = make_snowpack([0.3, 10], "independent_sphere", density=[150, 200], temperature=[260, 265], radius=[2e-4, 3e-4])
    solver = DORT(n_max_stream=8)

    class Anisotropic(Rayleigh):
        def ks(self, mu, npol=2):
            return np.outer(np.ones(npol), 1.0 + np.asarray(mu))

    class NotReciprocal(Rayleigh):
        def ft_even_phase(self, mu_s, mu_i, m_max, npol=None):
            P = super().ft_even_phase(mu_s, mu_i, m_max, npol)
            P[0, 0, 0] *= 1.0 + np.asarray(mu_i)[None, :]
            return P

    class Lopsided(Rayleigh):   # down-down block 2 % stronger than up-up: phase_symmetrization averages the two
        def ft_even_phase(self, mu_s, mu_i, m_max, npol=None):
            P = super().ft_even_phase(mu_s, mu_i, m_max, npol)
            h = len(mu_s) // 2
            P[..., h:, h:] *= 1.02
            return P

    class Dense(Rayleigh):   # (a pass-through ft_even_phase of its own keeps the class on the dense route: the Rayleigh
        def ft_even_phase(self, mu_s, mu_i, m_max, npol=None):   # family itself hands over its scalars only, below)
            return super().ft_even_phase(mu_s, mu_i, m_max, npol)

    plain = solver._pack(passive(37e9, 55), [sp2], np.array([37e9]), [[(Dense, {})] * 2])
    # the family itself: kind SMRT_EM_RAYLEIGH_HOST, ks / ka / permittivity only, the same numbers
    fam = solver._pack(passive(37e9, 55), [sp2], np.array([37e9]), [[(Rayleigh, {})] * 2])
    assert (fam.layer_kind & 15 == 7).all() and not fam.struct.host_phase and not fam.struct.host_streams
    np.testing.assert_array_equal(fam.host_layer, plain.host_layer)
    lop = DORT(n_max_stream=8, phase_symmetrization=True)._pack(passive(37e9, 55), [sp2], np.array([37e9]), [[(Lopsided, {})] * 2])
    np.testing.assert_allclose(lop.host_phase[..., 0, :, :], 1.01 * plain.host_phase[..., 0, :, :], rtol=1e-12)
    np.testing.assert_allclose(lop.host_phase[..., 1, :, :], plain.host_phase[..., 1, :, :], rtol=1e-12)

    for cls, msg in ((Anisotropic, "isotropic"), (NotReciprocal, "reciprocity")):
        with pytest.raises(SMRTError, match=msg):
            solver._pack(passive(37e9, 55), [sp2], np.array([37e9]), [[(cls, {})] * 2])
    # a device emmodel on a microstructure it does not know is refused before anything is launched
    homog = make_snowpack([0.3, 10], "homogeneous", density=[150, 200], temperature=[260, 265])
    with pytest.raises(SMRTError, match="no device implementation"):
        solver._pack(passive(37e9, 55), [homog], np.array([37e9]), "iba")
    # (independent spheres are a device microstructure model since round 4: IBA on them packs like any other)
    assert int(solver._pack(passive(37e9, 55), [sp2], np.array([37e9]), "iba").struct.microstructure) == 2


def test_host_evaluated_substrates_matrices_and_refusals():
    """DORT.substrate_matrices (no GPU needed): from what a substrate object returns -- specular diagonal, raw diffuse modes,
    dense (geometrical optics) or diagonal in the streams (IEM) -- to the reflection matrices of the bottom boundary, equal to
    the ones the reference built (stored in the fixtures); such a substrate is accepted by Snowpack, refused in passive mode."""
    from conftest import ROUGH_SUBSTRATE_FIXTURES, load_golden, model_snowpack_from_fixture
    from smrt_amd.core.snowpack import Snowpack, substrate_kind
    from smrt_amd.rtsolver.dort import DORT

    for name in ROUGH_SUBSTRATE_FIXTURES:
        d = load_golden(name)

        class FromFixture:
            temperature = 268.0

            def specular_reflection_matrix(self, frequency, eps_1, mu1, npol):
                np.testing.assert_allclose(mu1, d["sub_mu"], rtol=1e-12)
                return d["sub_spec_raw"]

            def ft_even_diffuse_reflection_matrix(self, frequency, eps_1, mu_s, mu_i, m_max, npol):
                return d["sub_diff_raw"]

        dense, coh = DORT.substrate_matrices(FromFixture(), float(d["frequency"][0]), 1.5, d["sub_mu"], d["sub_weight"], 2)
        for m in range(3):
            np.testing.assert_allclose(dense[m], d["sub_R_m%d" % m], rtol=1e-13, atol=1e-18)
            np.testing.assert_allclose(coh[m], d["sub_Rcoh_m%d" % m], rtol=1e-13, atol=1e-18)
        assert substrate_kind(FromFixture()) == "host"
        sp = model_snowpack_from_fixture(d)
        assert Snowpack(layers=sp.layers, substrate=FromFixture()).substrate is not None
    # passive mode: two polarisations, one mode; the matrices of the fixtures the reference could run
    from conftest import ROUGH_SUBSTRATE_PASSIVE_FIXTURES
    for name in ROUGH_SUBSTRATE_PASSIVE_FIXTURES:
        d = load_golden(name)

        class Passive:
            def specular_reflection_matrix(self, frequency, eps_1, mu1, npol):
                assert npol == 2
                return d["sub_spec_raw"]

            def ft_even_diffuse_reflection_matrix(self, frequency, eps_1, mu_s, mu_i, m_max, npol):
                assert npol == 2 and m_max == 0
                return d["sub_diff_raw"]

        dense, _ = DORT.substrate_matrices(Passive(), float(d["frequency"][0]), 1.5, d["sub_mu"], d["sub_weight"], 0, 2)
        np.testing.assert_allclose(dense[0], d["sub_R_m0"], rtol=1e-13, atol=1e-18)
    with pytest.raises(SMRTError, match="protocol"):
        Snowpack(layers=sp.layers, substrate=object())


def test_layer_density_is_read_only_and_update_keeps_the_layer_consistent():
    """`layer.density = x` used to be accepted and ignored (frac_volume lives on the microstructure object).
    Like the reference (smrt/inputs/make_medium.py:355-359) the three attributes frac_volume derives from are read-only
    once the layer exists; update(density=...) recomputes the ice volume fraction and the snowpack's packed columns."""
    from smrt_amd.core.error import SMRTError

    sp = two_layer()
    before = sp.packed().copy()
    for attr in ("density", "liquid_water", "volumetric_liquid_water"):
        with pytest.raises(SMRTError, match="read-only"):
            setattr(sp.layers[0], attr, 400.0)
    assert np.array_equal(sp.packed(), before)
    sp.layers[0].update(density=400.0)
    assert sp.layers[0].density == 400.0 and abs(sp.layers[0].frac_volume - 400.0 / 916.7) < 1e-12
    assert abs(sp.packed()[1, 0] - 400.0 / 916.7) < 1e-12 and sp.packed()[1, 0] != before[1, 0]
    sp.layers[0].update(liquid_water=0.1)          # wet snow (round 4): frac_volume becomes ice + water
    assert sp.layers[0].liquid_water == 0.1 and abs(sp.layers[0].frac_volume - 400.0 / (916.7 * 0.9 + 100.0)) < 1e-12
    sp.layers[0].update(liquid_water=0)
    assert abs(sp.layers[0].frac_volume - 400.0 / 916.7) < 1e-12 and sp.liquid_water() is None
    # the caches of one snowpack do not care about layers built or changed elsewhere
    filled = sp.packed()
    other = two_layer()
    other.layers[0].temperature = 200.0
    assert sp.packed() is filled


def test_emmodel_options_forms_and_diagonalization_warning():
    """The three forms of emmodel options of smrt/core/model.py:556-569 (one dict, a sequence with one dict per layer, a
    dict of dicts keyed by the medium next to a dict of emmodels) are resolved per layer or refused -- never mis-applied;
    a diagonalization_method other than the default is announced as ignored."""
    import warnings

    from smrt_amd import make_model
    from smrt_amd.core.error import SMRTError, SMRTWarning
    from smrt_amd.rtsolver.dort import DORT

    sp = two_layer()
    m = make_model("iba", "dort", emmodel_options=[dict(dense_snow_correction=None), {}])
    assert m.emmodel_options_of_layer(sp.layers[0], 0, 2) == dict(dense_snow_correction=None)
    assert m.emmodel_options_of_layer(sp.layers[1], 1, 2) == {}
    with pytest.raises(SMRTError, match="same length"):
        m.emmodel_options_of_layer(sp.layers[0], 0, 3)
    with pytest.raises(SMRTError, match="Mapping"):
        make_model("iba", "dort", emmodel_options=[1, 2])
    md = make_model({"snow": "iba"}, "dort", emmodel_options={"snow": dict(dense_snow_correction=None)})
    assert md.emmodel_options_of_layer(sp.layers[0], 0, 2) == dict(dense_snow_correction=None)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        DORT(diagonalization_method="stamnes88")
        DORT()
    assert len(w) == 1 and issubclass(w[0].category, SMRTWarning) and "stamnes88" in str(w[0].message)


@pytest.mark.parametrize("name", __import__("conftest").ROUGH_INTERFACE_FIXTURES)
def test_host_evaluated_interfaces_matrices(name):
    """DORT.interface_matrices (no GPU needed): from what an interface object with the reference's protocol returns on the
    stream grids -- here the replayed outputs of the reference's own iem_fung92 / geometrical_optics objects -- to the four
    combined matrices per azimuth mode that compute_interface_properties builds (rtsolver_utils.py:473-642): equal to the
    reference's matrices stored in the fixture; and the streams placed on the host equal the reference's."""
    from conftest import ReplayInterface, fixture_interfaces, load_golden
    from smrt_amd.rtsolver.dort import DORT

    d = load_golden(name)
    act = str(d["mode"]) == "A"
    i = int(d["rough_interface"][0])
    m_max, npol = (int(d["opt_m_max"]), 3) if act else (0, 2)
    eps = d["f0_effective_permittivity"]
    mus, ws, outmu, outw = DORT._streams_of(eps, int(d["opt_n_max_stream"]))
    assert [len(m) for m in mus] == list(d["streams_n"])
    np.testing.assert_allclose(mus[i], d["itf_mu_low"], rtol=1e-13)
    np.testing.assert_allclose(ws[i], d["itf_w_low"], rtol=1e-12)
    mu_up, w_up = (mus[i - 1], ws[i - 1]) if i > 0 else (outmu, outw)
    np.testing.assert_allclose(mu_up, d["itf_mu_up"], rtol=1e-13)
    np.testing.assert_allclose(w_up, d["itf_w_up"], rtol=1e-12)
    modes, coh = DORT.interface_matrices(ReplayInterface(d), float(d["frequency"][0]), complex(d["itf_eps_low"][0]),
                                         complex(d["itf_eps_up"][0]), d["itf_mu_low"], d["itf_mu_up"], d["itf_mu_t_up"],
                                         d["itf_w_low"], d["itf_w_up"], m_max, npol)
    want = fixture_interfaces(d)[i]
    for m in range(m_max + 1):
        for kind in ("Rtop", "Ttop", "Rbot", "Tbot"):
            A, B = modes[m][kind], np.asarray(want[kind][m])
            assert A.shape == B.shape, (kind, m, A.shape, B.shape)
            np.testing.assert_allclose(A, B, rtol=1e-13, atol=1e-300)
    for kind in ("Rtop", "Ttop", "Rbot", "Tbot"):
        np.testing.assert_allclose(coh[kind], np.diag(np.asarray(want[kind + "_coh"][0])), rtol=1e-13, atol=1e-300)


def test_snowpack_caches_follow_layer_changes():
    """The batching path caches per snowpack what it needs for every run (packed layer columns, microstructure set,
    per-layer emmodel flag); changing a layer afterwards -- a temperature for a sensitivity study, a correlation length set
    on the layer, a layer swapped in the list, an emmodel given to one layer -- must be seen by the next run."""
    sp = two_layer()
    a = sp.packed().copy()
    assert not sp.has_layer_emmodels() and sp.microstructure_models == {"exponential"}
    sp.layers[0].temperature = 240.0
    assert sp.packed()[2, 0] == 240.0 and a[2, 0] != 240.0
    sp.layers[1].corr_length = 3.3e-4                       # forwarded to the microstructure object the solver reads
    assert sp.packed()[3, 1] == 3.3e-4 and sp.layers[1].microstructure.corr_length == 3.3e-4
    sp.layers[0].emmodel = "nonscattering"
    assert sp.has_layer_emmodels()
    other = two_layer().layers[0]
    other.thickness = 0.77
    sp.packed()
    sp.layers[0] = other                                     # same count, another object
    assert sp.packed()[0, 0] == 0.77 and not sp.has_layer_emmodels()


def _own_objects_of(case):
    """smrt_amd's own make_snowpack / sensor objects for the numbers of a reference_objects.json case."""
    from conftest import _undump
    from smrt_amd import make_snowpack
    from smrt_amd.atmosphere.simple_isotropic_atmosphere import SimpleIsotropicAtmosphere
    from smrt_amd.core.sensor import Sensor
    from smrt_amd.substrate.flat import Flat
    from smrt_amd.substrate.reflector import Reflector

    packs = []
    for d in case["snowpacks"]:
        lay = d["layers"]
        ms = lay[0]["microstructure"]["cls"][1]
        kw = dict(density=[x["attrs"]["density"] for x in lay], temperature=[x["attrs"]["temperature"] for x in lay])
        if ms == "Exponential":
            name, kw["corr_length"] = "exponential", [x["microstructure"]["attrs"]["corr_length"] for x in lay]
        else:
            name = "sticky_hard_spheres"
            kw["radius"] = [x["microstructure"]["attrs"]["radius"] for x in lay]
            kw["stickiness"] = [x["microstructure"]["attrs"]["stickiness"] for x in lay]
        sub = d["substrate"]
        if sub is not None and sub["cls"][1] == "Flat":
            table = {f: _undump(e) for f, e in sub["permittivity"]}
            kw["substrate"] = Flat(temperature=sub["temperature"], permittivity_model=lambda f, _t=None, table=table: table[f])
        elif sub is not None:
            kw["substrate"] = Reflector(temperature=sub["temperature"], specular_reflection=_undump(sub["specular_reflection"]))
        if d["atmosphere"] is not None:
            a = {k: _undump(v) for k, v in d["atmosphere"]["attrs"].items()}
            kw["atmosphere"] = SimpleIsotropicAtmosphere(tb_down=a["constant_tbdown"], tb_up=a["constant_tbup"],
                                                         transmittance=a["constant_trans"])
        packs.append(make_snowpack([x["attrs"]["thickness"] for x in lay], name, **kw))
    sensors = []
    for s in case["sensors"]:
        a = s["attrs"]
        sensors.append(Sensor(frequency=a["frequency"], theta_deg=a["theta_deg"], theta_inc_deg=a["theta_inc_deg"],
                              polarization=a["polarization"], polarization_inc=a["polarization_inc"],
                              phi_deg=None if not np.any(a["phi"]) else np.rad2deg(a["phi"]), channel_map=_undump(a["channel_map"])))
    return [(sensors[i], packs[j]) for i, j in case["simulations"]]


class _RecordingContext:
    """Stands where rtsolver/dort.py:get_context returns the GPU context: records the batches, answers zeros."""

    def __init__(self):
        import threading

        self.lock, self.batches = threading.RLock(), []

    def set_block_threads(self, n):
        pass

    def run(self, batch, lo=0, n=None, pairs=None):
        from smrt_amd._native import BatchOutput

        self.batches.append((batch, None if pairs is None else np.array(pairs)))
        out = BatchOutput(batch, (batch.n_pairs - lo if n is None else n) if pairs is None else len(pairs))
        for a in (out.values, out.status, out.layers, out.streams):
            a[...] = 0
        out.streams[:, :3] = 2.0, 0.9, 0.5      # two air streams, so that the active result finds its incident ones
        return out


def _batch_bytes(batch):
    import ctypes as C

    out = {k: (v.dtype.str, v.shape, v.tobytes()) for k, v in vars(batch).items() if isinstance(v, np.ndarray)}
    for k, v in vars(batch).items():
        if isinstance(v, list) and v and all(isinstance(a, np.ndarray) for a in v):
            out[k] = [(a.dtype.str, a.shape, a.tobytes()) for a in v]
    for name, ctype in batch.struct._fields_:
        if ctype in (C.c_int32, C.c_double):
            out["struct." + name] = getattr(batch.struct, name)
    return out


@pytest.mark.parametrize("name", ["headline_shape_two_snowpacks", "flat_substrate_and_atmosphere", "dmrt_on_a_reflector",
                                  "active_dense_auto"])
def test_reference_shaped_objects_are_packed_like_own_objects(name, monkeypatch):
    """The runner protocol fed with stand-ins that carry the class identities and public attributes of the REFERENCE's
    Model / Sensor / Snowpack / Layer objects (tests/golden/reference_objects.json, dumped from real ones) -- what
    smrt/core/model.py:395-398 hands over -- gives bitwise the device batch smrt_amd's own objects give, with the
    reference's emmodel CLASS mapped to the device emmodel (incl. dense_snow_correction="auto" layer by layer).  The
    executed counterpart with the real package: tests/test_reference_binding.py; on the GPU: tests/test_gpu_model.py."""
    import smrt_amd.rtsolver.dort as dort
    from conftest import load_reference_objects, standins_from_dump
    from smrt_amd.rtsolver.dort import DORT
    from smrt_amd.runner.hip_batch_runner import HipBatchRunner

    ctx = _RecordingContext()
    monkeypatch.setattr(dort, "get_context", lambda device=None: ctx)
    case = load_reference_objects()[name]
    model, sims, packs, _ = standins_from_dump(case)
    results = HipBatchRunner()(model.run_single_simulation, [(sim, None, "outer") for sim in sims])
    assert len(results) == len(sims)
    foreign = list(ctx.batches)
    del ctx.batches[:]
    own = _own_objects_of(case)
    emmodel = {"IBA": "iba", "DMRT_QCA_ShortRange": "dmrt_qca_shortrange"}[case["model"]["emmodel"][1]]
    options = model.rtsolver_options
    if name == "active_dense_auto":   # what smrt_amd's own Model does with that option: per-layer device names
        entries = [["iba_inverted" if lay.frac_volume > 0.5 else "iba" for lay in own[0][1].layers]]
        DORT(**options).solve_batch(own, entries)
    else:
        DORT(**options).solve_batch(own, emmodel)
    assert len(foreign) == len(ctx.batches) == 1
    a, b = _batch_bytes(foreign[0][0]), _batch_bytes(ctx.batches[0][0])
    assert a.keys() == b.keys()
    for k in a:
        assert a[k] == b[k], f"the device batches differ in '{k}'"
    assert (foreign[0][1] is None) == (ctx.batches[0][1] is None)


def test_layers_of_the_reference_the_device_cannot_compute_are_not_computed_as_dry_snow():
    """core/foreign.py: a scatterer permittivity that is not the default one, non-spherical inclusions, another microstructure
    model -> the layer carries the reason (wet snow with the default model is the device's business); a device descriptor on it raises, a reference class on it is kept for the host route."""
    from conftest import load_reference_objects, standin_class, standin_function, standins_from_dump
    from smrt_amd.core.error import SMRTError
    from smrt_amd.core.foreign import adopt_snowpack, device_entry, entry_of_instance
    from smrt_amd.emmodel.iba import IBA

    case = load_reference_objects()["flat_substrate_and_atmosphere"]
    model, sims, packs, _ = standins_from_dump(case)
    sp = packs[0]
    ok = adopt_snowpack(sp).layers
    assert [lay.device_refusal for lay in ok] == [None] * 3
    ref_iba = model.emmodel
    assert device_entry(ref_iba, {}, ok[0]) == "iba" and device_entry(IBA, {}, ok[0]) == "iba"
    assert device_entry(ref_iba, {"dense_snow_correction": "auto"}, ok[0]) == "iba"
    specialised = type("Specialized IBA", (ref_iba,), {"__module__": ref_iba.__module__})
    assert device_entry(specialised, {}, ok[0]) == (specialised, {})          # (a subclass may compute anything)
    sp.layers[0].liquid_water = 0.02           # wet snow with the default permittivity model runs on the device ...
    wet = adopt_snowpack(sp)
    assert wet.layers[0].device_refusal is None and wet.liquid_water()[0] == 0.02 and device_entry(ref_iba, {}, wet.layers[0]) == "iba"
    dry_ice = sp.layers[0].permittivity_model[1]
    sp.layers[0].permittivity_model = (1.0, standin_function("smrt.permittivity.ice", "ice_permittivity_maetzler06"))
    sp.layers[1].inclusion_shape = "random_needles"
    sp.layers[2].permittivity_model = (1.0, lambda f, **k: 3.2 + 0.001j)
    bad = adopt_snowpack(sp).layers               # ... but not with a scatterer permittivity that ignores the water
    assert "liquid water" in bad[0].device_refusal and "inclusion_shape" in bad[1].device_refusal
    assert "permittivity" in bad[2].device_refusal
    for lay in bad:
        assert device_entry(ref_iba, {}, lay) == (ref_iba, {})
        with pytest.raises(SMRTError, match="cannot compute this layer"):
            device_entry(IBA, {}, lay)
        inst = ref_iba(sims[0][0], lay.source)
        assert entry_of_instance(inst, lay) is inst
    sp.layers[0].liquid_water = 0
    sp.layers[0].permittivity_model = (1.0, dry_ice)
    sp.layers[0].microstructure = standin_class("smrt.microstructure_model.gaussian_random_field", "GaussianRandomField")()
    sp.layers[0].microstructure.frac_volume = 0.3
    assert "no device implementation" in adopt_snowpack(sp).layers[0].device_refusal


def test_make_snowpack_refusals_and_defaults_follow_the_reference():
    """surface + a sequence of interfaces is ambiguous (smrt/inputs/make_medium.py:207-210); sticky hard spheres default to
    stickiness 1000 (smrt/microstructure_model/sticky_hard_spheres.py:30)."""
    from smrt_amd import make_snowpack
    from smrt_amd.core.error import SMRTError
    from smrt_amd.interface.flat import Flat

    with pytest.raises(SMRTError, match="ambiguous"):
        make_snowpack([0.1, 1.0], "exponential", density=300, corr_length=1e-4, interface=[Flat, Flat], surface=Flat)
    sp = make_snowpack([0.1, 1.0], "exponential", density=300, corr_length=1e-4, interface=Flat, surface=Flat)
    assert sp.nlayer == 2
    shs = make_snowpack([1.0], "sticky_hard_spheres", density=300, radius=1e-4)
    assert shs.layers[0].microstructure.device_params == (1e-4, 1000.0) and shs.packed()[4, 0] == 1000.0


def _close(got, want, rtol=1e-12):
    got, want = np.asarray(got, float), np.asarray(want, float)
    if want.ndim == 0 or not np.any(want):
        assert not np.any(got)
        return
    np.testing.assert_allclose(got.reshape(want.shape), want, rtol=rtol, atol=rtol * np.abs(want).max())


def test_own_rough_interface_evaluators_against_the_reference_outputs():
    """smrt_amd/interface/iem_fung92.py and geometrical_optics.py (own NumPy evaluators of Fung et al. 1992 and of Tsang
    vol. III section 2.1) against what the reference's objects returned on the stream grids of the fixtures
    (itf_raw_*: specular reflection, coherent transmission, azimuth modes of the diffuse reflection / transmission, both
    sides of the interface): 1e-12."""
    import warnings

    from conftest import ROUGH_INTERFACE_MODELS, load_golden
    from smrt_amd import make_interface

    for name, (model, kw) in ROUGH_INTERFACE_MODELS.items():
        d = load_golden(name)
        itf = make_interface(model, **kw)
        f, act = float(d["frequency"][0]), str(d["mode"]) == "A"
        npol, m_max = (3, 2) if act else (2, 0)
        lo, up = complex(d["itf_eps_low"][0]), complex(d["itf_eps_up"][0])
        mu_l, mu_u, mu_t = d["itf_mu_low"], d["itf_mu_up"], d["itf_mu_t_up"]
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")      # (the IEM validity warning, like the reference's)
            _close(itf.specular_reflection_matrix(f, lo, up, mu_l, npol), d["itf_raw_spec_up"])
            _close(itf.specular_reflection_matrix(f, up, lo, mu_u, npol), d["itf_raw_spec_dn"])
            _close(itf.coherent_transmission_matrix(f, lo, up, mu_l, npol), d["itf_raw_ctr_up"])
            _close(itf.coherent_transmission_matrix(f, up, lo, mu_u, npol), d["itf_raw_ctr_dn"])
            _close(itf.ft_even_diffuse_reflection_matrix(f, lo, up, mu_l, mu_l, m_max, npol), d["itf_raw_drf_up"])
            _close(itf.ft_even_diffuse_reflection_matrix(f, up, lo, mu_u, mu_u, m_max, npol), d["itf_raw_drf_dn"])
            if "itf_raw_dtr_up" in d:
                _close(itf.ft_even_diffuse_transmission_matrix(f, lo, up, mu_t, mu_l, m_max, npol), d["itf_raw_dtr_up"])
                _close(itf.ft_even_diffuse_transmission_matrix(f, up, lo, mu_l, mu_u, m_max, npol), d["itf_raw_dtr_dn"])


def test_own_rough_substrates_against_the_reference_outputs():
    """make_soil("iem_fung92" | "geometrical_optics" | "geometrical_optics_backscatter", eps, T, ...): the substrate protocol
    on the streams of the last layer against what the reference's substrates returned (sub_*_raw), incl. the hemispherical
    integration behind the emissivity of the backscatter-only geometrical optics: 1e-12."""
    import warnings

    from conftest import ROUGH_SUBSTRATE_MODELS, load_golden
    from smrt_amd import make_soil
    from smrt_amd.core.snowpack import substrate_kind

    for name, (model, kw) in ROUGH_SUBSTRATE_MODELS.items():
        d = load_golden(name)
        soil = make_soil(model, complex(8.0, 1.0), 268.0, **kw)
        assert substrate_kind(soil) == "host" and soil.temperature == 268.0
        f, act = float(d["frequency"][0]), str(d["mode"]) == "A"
        npol, m_max = (3, 2) if act else (2, 0)
        mu, eps = d["sub_mu"], complex(d["f0_effective_permittivity"][len(d["thickness"]) - 1])
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            _close(soil.specular_reflection_matrix(f, eps, mu, npol), d["sub_spec_raw"])
            _close(soil.ft_even_diffuse_reflection_matrix(f, eps, mu, mu, m_max, npol), d["sub_diff_raw"])
            if "sub_emis_raw" in d:
                _close(soil.emissivity_matrix(f, eps, mu, npol), d["sub_emis_raw"])
    with pytest.raises(Exception, match="outside the scope"):
        make_soil("flat", "dobson85", 268.0)


def test_wet_snow_layers_follow_the_reference_definitions():
    """make_snowpack(volumetric_liquid_water=... | liquid_water=...): frac_volume = (ice + water) / total volume and
    liquid_water = water / (ice + water) as SnowLayer.compute_frac_volumes (smrt/inputs/make_medium.py:390-434; the
    fixture holds what the reference computed), read-only afterwards except through update(); the liquid-water column of
    the device batch; the scatterer permittivity of a wet layer on the host (wetice.py:12-45) against the oracle's."""
    from conftest import load_golden
    from oracle import dort_oracle as O
    from smrt_amd import make_snowpack
    from smrt_amd.core.error import SMRTError

    d = load_golden("iba_wet_L4_n12_passive")
    sp = make_snowpack(d["thickness"], "exponential", density=d["density"], temperature=d["temperature"],
                       corr_length=d["corr_length"], volumetric_liquid_water=[0.03, 0.005, 0.0, 0.0])
    np.testing.assert_allclose([lay.frac_volume for lay in sp.layers], d["frac_volume"], rtol=1e-14)
    np.testing.assert_allclose(sp.liquid_water(), d["liquid_water"], rtol=1e-14)
    again = make_snowpack(d["thickness"], "exponential", density=d["density"], temperature=d["temperature"],
                          corr_length=d["corr_length"], liquid_water=list(d["liquid_water"]))
    np.testing.assert_allclose([lay.frac_volume for lay in again.layers], d["frac_volume"], rtol=1e-13)
    dry = make_snowpack([1.0], "exponential", density=300, corr_length=1e-4)
    assert dry.liquid_water() is None and dry.layers[0].liquid_water == 0
    with pytest.raises(SMRTError, match="read-only"):
        sp.layers[0].liquid_water = 0.5
    with pytest.raises(SMRTError, match="ambiguous"):
        make_snowpack([1.0], "exponential", density=300, corr_length=1e-4, liquid_water=0.1, volumetric_liquid_water=0.02)
    sp.layers[2].update(volumetric_liquid_water=0.01)
    assert sp.liquid_water()[2] > 0
    eps = sp.layers[0].permittivity(1, 18.7e9)
    assert abs(eps - O.scatterer_permittivity(18.7e9, 273.15, float(d["liquid_water"][0]))) < 1e-12
    assert eps.imag > 10 * sp.layers[3].permittivity(1, 18.7e9).imag      # (water makes the grains lossy)


def test_unified_microstructure_parameters_are_reparametrisations_of_the_device_forms():
    """core/layer.py: device_microstructure_params -- the models on the unified parameters (porod length, polydispersity)
    expressed through the closed forms the device has (dort_physics.hpp: ft_corr): against the reference's own expressions
    (restated in the oracle) over the wavenumbers IBA samples, on both sides of polydispersity 1."""
    from oracle import dort_oracle as O
    from smrt_amd.core.layer import device_microstructure_params as dp

    k = np.linspace(0.0, 6e4, 601)
    for f, lp in ((0.15, 0.8e-4), (0.3, 1.2e-4), (0.45, 2.5e-4)):
        for K in (0.5, 0.8, 0.999, 1.0, 1.001, 1.3, 2.0, 3.5):
            xi, Y = dp("unified_teubner_strey", f, porod_length=lp, polydispersity=K)
            assert (Y >= 0) == (K <= 1.0) or abs(Y) < 1e-3
            x = (k * xi) ** 2
            device_form = f * (1 - f) * 8 * np.pi * xi**3 / ((1 + Y) ** 2 + 2 * (1 - Y) * x + x**2)
            np.testing.assert_allclose(device_form, O.ft_autocorr_unified_teubner_strey(k, f, lp, K), rtol=1e-11)
            lc, zero = dp("unified_scaled_exponential", f, porod_length=lp, polydispersity=K)
            assert zero == 0.0
            np.testing.assert_allclose(O.ft_autocorr_exponential(k, f, lc), O.ft_autocorr_unified_scaled_exponential(k, f, lp, K), rtol=1e-13)
            radius, minus_t = dp("unified_sticky_hard_spheres", f, porod_length=lp, polydispersity=K)
            assert minus_t < 0
            np.testing.assert_allclose(O.ft_autocorr_shs(k, f, radius, None, t=-minus_t), O.ft_autocorr_unified_shs(k, f, lp, K), rtol=1e-13)
    # t <= 0 (small polydispersity) has no device encoding -- the sign of micro_p2 tells "t given" from "stickiness given" --
    # and an object's own derived radius / t win over the ones recomputed from frac_volume (smrt's inverted_medium() copy)
    from smrt_amd.core.error import SMRTError
    with pytest.raises(SMRTError, match="no device encoding"):
        dp("unified_sticky_hard_spheres", 0.2, porod_length=1e-4, polydispersity=0.3)
    assert dp("unified_sticky_hard_spheres", 0.3, porod_length=1e-4, polydispersity=1.0, radius=5e-4, t=10.17) == (5e-4, -10.17)
    # Teubner-Strey itself: micro_p2 is Y = (2 pi xi / d)^2
    xi, Y = dp("teubner_strey", 0.3, corr_length=1.5e-4, repeat_distance=1.2e-3)
    assert xi == 1.5e-4 and Y == (2 * np.pi * 1.5e-4 / 1.2e-3) ** 2


def test_snowpack_caches_follow_in_place_changes():
    """Snowpack.liquid_water() / all_interfaces_flat() are looked up once per snowpack (the batching runner asks on every run
    of every snowpack): a layer made wet or dry with update(), a layer appended, an interface replaced in place must all
    show at the next call."""
    from smrt_amd import make_snowpack
    from smrt_amd.inputs.make_medium import make_interface, make_snow_layer

    sp = make_snowpack([0.2, 0.3, 10.0], "exponential", density=[250, 300, 350], temperature=[273.15, 265, 260], corr_length=1e-4)
    assert sp.liquid_water() is None and sp.all_interfaces_flat()
    sp.layers[0].update(volumetric_liquid_water=0.02)
    assert sp.liquid_water()[0] > 0 and sp.liquid_water()[1] == 0
    sp.layers[0].update(volumetric_liquid_water=0.0)
    assert sp.liquid_water() is None
    rough = make_interface("iem_fung92", roughness_rms=1e-3, corr_length=5e-2)
    sp.interfaces[1] = rough
    assert not sp.all_interfaces_flat()
    sp.interfaces[1] = type(sp.interfaces[0])()
    assert sp.all_interfaces_flat()
    sp.append(make_snow_layer(1.0, "exponential", density=400, temperature=273.15, corr_length=2e-4, volumetric_liquid_water=0.01), rough)
    assert sp.liquid_water()[3] > 0 and not sp.all_interfaces_flat()


def test_packed_columns_of_a_repeated_run_follow_changed_snowpacks(emulated):
    """DORT._pack keeps the stacked columns of the last group ON THE MODEL of the run and copies them when a run comes
    with the very same, unchanged snowpacks (Model._kept_columns): the next batch must be the same bytes, a layer changed in
    place or a snowpack swapped must show in it, and the rows of the others must not move."""
    from smrt_amd.runner.hip_batch_runner import HipBatchRunner

    emulated.compute = False   # host side only: the batches as packed
    rng = np.random.default_rng(5)
    S = 300
    sps = [make_snowpack([0.1, 0.3, 20.0], "exponential", density=rng.uniform(200, 400, 3), temperature=rng.uniform(240, 270, 3),
                         corr_length=rng.uniform(5e-5, 3e-4, 3)) for _ in range(S)]
    sensor = sensor_list.passive([19e9, 37e9], 55.0)
    m = make_model("iba", "dort", rtsolver_options=dict(n_max_stream=8))
    assert m._kept_columns[1] is None
    cols = ("thickness", "frac_volume", "temperature", "micro_p1", "micro_p2")

    def packed():
        del emulated.batches[:]
        m.run(sensor, sps, runner=HipBatchRunner())
        b = emulated.batches[-1][0]
        return {c: np.array(getattr(b, c), copy=True) for c in cols}

    first = packed()
    assert m._kept_columns[1] is not None and len(m._kept_columns[1][0]) == S
    assert make_model("iba", "dort")._kept_columns[1] is None      # (another model has its own slot)
    again = packed()                       # served from the kept columns
    assert all(first[c].tobytes() == again[c].tobytes() for c in cols)
    sps[7].layers[1].update(temperature=251.25)          # a layer changed in place
    sps[200] = make_snowpack([0.5, 20.0, 30.0], "exponential", density=[310, 320, 330], temperature=[255, 256, 257], corr_length=1e-4)
    third = packed()
    assert third["temperature"][7, 1] == 251.25 and first["temperature"][7, 1] != 251.25
    assert np.array_equal(third["thickness"][200], [0.5, 20.0, 30.0])
    keep = np.ones(S, bool); keep[[7, 200]] = False
    assert all(np.array_equal(third[c][keep], first[c][keep]) for c in cols)
    m._kept_columns[1] = None
    fresh = packed()                       # packed from scratch: the same bytes as with the kept columns
    assert all(fresh[c].tobytes() == third[c].tobytes() for c in cols)


def test_a_layer_changed_after_a_run_shows_in_the_next_batch_of_every_route(emulated):
    """Model.run used to leave its per-run snapshot of the layer columns ON the Snowpack, and the
    per-simulation routes (DORT.solve_batch, the runner protocol) then packed the stale tuple.  The snapshot now lives on
    the solver for the duration of one solve: after a run, a changed layer -- or an appended one -- is what every route
    packs, and no route leaves anything but its own caches on the snowpack."""
    from smrt_amd.rtsolver.dort import DORT
    from smrt_amd.runner.hip_batch_runner import HipBatchRunner

    emulated.compute = False
    sp = make_snowpack([0.1, 20.0], "exponential", density=[250, 350], temperature=[260.0, 265.0], corr_length=1e-4)
    sensor = sensor_list.passive([19e9], 55.0)
    m = make_model("iba", "dort", rtsolver_options=dict(n_max_stream=8))
    m.run(sensor, [sp], runner=HipBatchRunner())
    assert "_f" not in sp.__dict__
    sp.layers[0].temperature = 200.0
    batch = DORT(n_max_stream=8)._pack(sensor, [sp], np.array([19e9]), "iba")
    assert list(np.asarray(batch.temperature).reshape(-1)[:2]) == [200.0, 265.0]
    del emulated.batches[:]
    DORT(n_max_stream=8).solve_batch([(sensor, sp)], "iba")
    assert np.asarray(emulated.batches[-1][0].temperature).reshape(-1)[0] == 200.0
    from smrt_amd.inputs.make_medium import make_snow_layer
    sp.append(make_snow_layer(0.5, "exponential", density=300, temperature=250.0, corr_length=2e-4))
    del emulated.batches[:]
    m.run(sensor, [sp], runner=HipBatchRunner())
    b = emulated.batches[-1][0]
    assert int(np.asarray(b.n_layers)[0]) == 3 and np.asarray(b.temperature).reshape(-1)[2] == 250.0


def test_two_models_pack_on_two_threads(emulated):
    """The kept columns are per Model and behind its lock (the process-wide slot raced): two models
    packing different ensembles on two threads, again and again, always get their own rows."""
    import threading

    from smrt_amd.runner.hip_batch_runner import HipBatchRunner

    emulated.compute = False
    rng = np.random.default_rng(9)
    S = 300
    sensor = sensor_list.passive([19e9], 55.0)
    ens, models, errors = [], [], []
    for k in range(2):
        ens.append([make_snowpack([0.1, 20.0], "exponential", density=rng.uniform(200, 400, 2), temperature=[240.0 + k, 250.0 + k],
                                  corr_length=rng.uniform(5e-5, 3e-4, 2)) for _ in range(S)])
        models.append(make_model("iba", "dort", rtsolver_options=dict(n_max_stream=8)))

    def work(k):
        try:
            from smrt_amd.rtsolver.dort import DORT
            for _ in range(12):
                r = DORT(n_max_stream=8)
                r._plan_model = models[k]
                b = r._pack(sensor, ens[k], np.array([19e9]), "iba")
                t = np.asarray(b.temperature).reshape(S, 2)
                if not (np.all(t[:, 0] == 240.0 + k) and np.all(t[:, 1] == 250.0 + k)):
                    errors.append(k)
        except Exception as e:   # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    assert all(m._kept_columns[1] is not None and len(m._kept_columns[1][0]) == S for m in models)


def test_stacked_diagnostics_are_those_of_the_solve_not_of_later_edits(emulated):
    """Result.other_data of a stacked result is built on first access -- from the layer counts and
    thickness rows snapshotted AT SOLVE TIME (rtsolver/dort.py: _Solution.columns), not from the live Snowpack objects, which
    a sensitivity loop changes between the run and the read (the reference builds other_data inside the solve,
    smrt/rtsolver/rtsolver_utils.py:322-344)."""
    from smrt_amd.inputs.make_medium import make_snow_layer
    from smrt_amd.runner.hip_batch_runner import HipBatchRunner

    sps = [make_snowpack([0.1, 0.3, 20.0], "exponential", density=[250, 300, 350], temperature=260.0, corr_length=1e-4),
           make_snowpack([0.2, 10.0], "exponential", density=[250, 350], temperature=255.0, corr_length=2e-4)]
    sensor = sensor_list.passive([19e9, 37e9], 55.0)
    res = make_model("iba", "dort", rtsolver_options=dict(n_max_stream=8)).run(sensor, sps, runner=HipBatchRunner())
    sps[0].layers[1].thickness = 7.0                       # edits AFTER the run, BEFORE the diagnostics are read
    sps[1].append(make_snow_layer(0.5, "exponential", density=300, temperature=250.0, corr_length=2e-4))
    th = np.asarray(res.other_data["thickness"].values)
    assert th.shape[-1] == 3
    assert np.array_equal(th[0, 0], [0.1, 0.3, 20.0]) and np.array_equal(th[1, 0], [0.1, 0.3, 20.0])
    assert np.array_equal(th[0, 1, :2], [0.2, 10.0]) and np.isnan(th[0, 1, 2])
