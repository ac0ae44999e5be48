"""The plugin surface on the GPU: make_model()/Model.run()/Result against the reference's known answers."""
import numpy as np
import pytest

from conftest import assert_backscatter_close, load_golden, reference_method_spread

pytestmark = pytest.mark.gpu


def two_layer():
    from smrt_amd import make_snowpack

    return make_snowpack([0.1, 100], "exponential", density=[200, 400], temperature=[250.0, 250.0],
                         corr_length=[5e-5, 5e-5])


def test_iba_dort_oneconfig_passive():
    """smrt/test/test_integration_iba.py:33-49, every diagonalization_method name of the reference."""
    from smrt_amd import make_model, sensor_list

    for method in ("eig", "schur", "half_rank_eig", "schur_forcedtriu"):
        m = make_model("iba", "dort", rtsolver_options=dict(diagonalization_method=method))
        res = m.run(sensor_list.amsre("37V"), two_layer())
        np.testing.assert_allclose(res.TbV(), 248.09044325849692, atol=1e-4)
        np.testing.assert_allclose(res.TbH(), 237.3487270223389, atol=1e-4)


def test_iba_dort_oneconfig_active():
    """smrt/test/test_integration_iba.py:55-69: the reference's own known answer for the radar case."""
    from smrt_amd import make_model, sensor_list

    res = make_model("iba", "dort").run(sensor_list.active(frequency=19e9, theta_inc=55), two_layer())
    np.testing.assert_allclose(res.sigmaVV_dB(), -24.044882546524693, atol=1e-4)
    np.testing.assert_allclose(res.sigmaHH_dB(), -24.416295329469907, atol=1e-4)
    np.testing.assert_allclose(res.sigmaHV_dB(), -51.544272924876886, atol=1e-4)
    d = load_golden("iba_2layer_active19")
    np.testing.assert_allclose(res.other_data["stream_angles"].values, d["f0_stream_angles"], rtol=1e-11)


def test_active_model_run_sentinel1_batch():
    """Model.run in active mode over several snowpacks: dims, labels and values of the cfg4-like fixture."""
    from smrt_amd import make_model, make_snowpack, sensor_list

    d = load_golden("cfg4_iba_active_L5_n16")
    sp = make_snowpack(d["thickness"], "exponential", density=d["density"], temperature=d["temperature"],
                       corr_length=d["corr_length"])
    m = make_model("iba", "dort", rtsolver_options=dict(n_max_stream=16, m_max=2))
    res = m.run(sensor_list.sentinel1(), [sp, sp, sp])
    assert "snowpack" in res.data.dims
    r0 = d["result"][0]
    th = d["theta_inc_deg"]
    want_vv = 10 * np.log10(4 * np.pi * np.cos(np.deg2rad(th)) * r0[0, 0, :])
    got = np.asarray(res.sigmaVV_dB().values if hasattr(res.sigmaVV_dB(), "values") else res.sigmaVV_dB())
    assert got.shape == (3, len(th))
    np.testing.assert_allclose(got, np.tile(want_vv, (3, 1)), atol=1e-7)


def test_substrate_and_atmosphere_through_the_model():
    """make_snowpack(substrate=, atmosphere=) / snowpack + substrate / atmosphere + snowpack, against the reference
    fixture; Kirchhoff check: a non-scattering isothermal pack over an isothermal substrate under a black sky at the
    same temperature radiates that temperature."""
    from smrt_amd import make_atmosphere, make_model, make_snowpack, sensor_list
    from smrt_amd.substrate.flat import Flat

    d = load_golden("iba_L3_n16_substrate_atmosphere")
    sub = Flat(temperature=float(d["substrate_temperature"]), permittivity_model=complex(d["substrate_eps"][0]))
    atm = make_atmosphere("simple_isotropic_atmosphere", tb_down=dict(zip(d["frequency"], d["atm_tb_down"])),
                          tb_up=dict(zip(d["frequency"], d["atm_tb_up"])),
                          transmittance=dict(zip(d["frequency"], d["atm_trans"])))
    kw = dict(density=d["density"], temperature=d["temperature"], corr_length=d["corr_length"])
    sp1 = make_snowpack(d["thickness"], "exponential", substrate=sub, atmosphere=atm, **kw)
    sp2 = atm + (make_snowpack(d["thickness"], "exponential", **kw) + sub)
    m = make_model("iba", "dort", rtsolver_options=dict(n_max_stream=16))
    sensor = sensor_list.passive(list(d["frequency"]), list(d["theta_deg"]))
    for sp in (sp1, sp2):
        res = m.run(sensor, sp)
        for i, f in enumerate(d["frequency"]):
            np.testing.assert_allclose(res.TbV(frequency=f), d["result"][i, 0], atol=1e-6)
            np.testing.assert_allclose(res.TbH(frequency=f), d["result"][i, 1], atol=1e-6)
    T = 265.0
    iso = make_snowpack([0.3, 0.5], "exponential", density=[250, 350], temperature=T, corr_length=1e-7,
                        substrate=Flat(temperature=T, permittivity_model=5 + 0.5j),
                        atmosphere=make_atmosphere(tb_down=T, tb_up=0.0, transmittance=1.0))
    res = m.run(sensor_list.passive(18.7e9, [30, 55]), iso)
    np.testing.assert_allclose(np.ravel(res.TbV()), T, atol=1e-3)
    np.testing.assert_allclose(np.ravel(res.TbH()), T, atol=1e-3)


def test_prune_deep_snowpack_through_the_model():
    """rtsolver_options=dict(prune_deep_snowpack=...) like in the reference (smrt/rtsolver/dort.py:117-124), against a
    fixture the reference produced with the option; without it the same pack is up to 48 K away."""
    from smrt_amd import make_model, make_snowpack, sensor_list
    from smrt_amd.substrate.flat import Flat

    d = load_golden("iba_L6_n16_prune_substrate")
    sub = Flat(temperature=float(d["substrate_temperature"]), permittivity_model=complex(d["substrate_eps"][0]))
    sp = make_snowpack(d["thickness"], "exponential", density=d["density"], temperature=d["temperature"],
                       corr_length=d["corr_length"], substrate=sub)
    sensor = sensor_list.passive(list(d["frequency"]), list(d["theta_deg"]))
    opts = dict(n_max_stream=16, prune_deep_snowpack=float(d["opt_prune_deep_snowpack"]))
    res = make_model("iba", "dort", rtsolver_options=opts).run(sensor, sp)
    full = make_model("iba", "dort", rtsolver_options=dict(n_max_stream=16)).run(sensor, sp)
    for i, f in enumerate(d["frequency"]):
        np.testing.assert_allclose(res.TbV(frequency=f), d["result"][i, 0], atol=1e-6)
        np.testing.assert_allclose(res.TbH(frequency=f), d["result"][i, 1], atol=1e-6)
    assert np.abs(np.ravel(full.TbV(frequency=36.5e9)) - d["result"][1, 0]).max() > 10.0


def test_other_emmodels_through_the_model():
    """dmrt_qcacp_shortrange (the reference's known answer, smrt/test/test_dmrtdort.py:20-37) and nonscattering over a
    Flat substrate through make_model()/Model.run()."""
    from smrt_amd import make_model, make_snowpack, sensor_list
    from smrt_amd.substrate.flat import Flat

    sp = make_snowpack([0.1, 1000], "sticky_hard_spheres", density=[200, 400], temperature=[250.0, 250.0],
                       radius=[2e-4, 2e-4], stickiness=[0.1, 0.1])
    res = make_model("dmrt_qcacp_shortrange", "dort").run(sensor_list.amsre("37V"), sp)
    assert abs(res.TbV() - 201.83572222) < 1e-6 and abs(res.TbH() - 187.29558162) < 1e-6
    d = load_golden("nonscattering_L3_n10_substrate")
    sp = make_snowpack(d["thickness"], "exponential", density=d["density"], temperature=d["temperature"],
                       corr_length=d["corr_length"],
                       substrate=Flat(temperature=float(d["substrate_temperature"]),
                                      permittivity_model=complex(d["substrate_eps"][0])))
    res = make_model("nonscattering", "dort", rtsolver_options=dict(n_max_stream=10)).run(
        sensor_list.passive(list(d["frequency"]), list(d["theta_deg"])), sp)
    for i, f in enumerate(d["frequency"]):
        np.testing.assert_allclose(res.TbV(frequency=f), d["result"][i, 0], atol=1e-6)
        np.testing.assert_allclose(res.TbH(frequency=f), d["result"][i, 1], atol=1e-6)


def test_onelayer_example():
    """examples/iba_onelayer_example.py."""
    from smrt_amd import make_model, make_snowpack, sensor_list

    sp = make_snowpack(thickness=[100], microstructure_model="exponential", density=[320], temperature=[270],
                       corr_length=[5e-5])
    res = make_model("iba", "dort").run(sensor_list.amsre("37V"), sp)
    assert abs(res.TbV() - 268.22172695) < 1e-6 and abs(res.TbH() - 251.75293753) < 1e-6
    d = load_golden("cfg1_iba_onelayer")
    np.testing.assert_allclose(res.other_data["ks"].values, d["f0_ks"], rtol=1e-11)
    np.testing.assert_allclose(res.other_data["ka"].values, d["f0_ka"], rtol=1e-10)
    np.testing.assert_allclose(res.other_data["stream_angles"].values, d["f0_stream_angles"], rtol=1e-11)
    assert res.optical_depth().values[0] > 5


def test_model_run_many_snowpacks_and_channels():
    """Model.run over snowpacks x AMSR-E frequencies: dims, labels, channel selection, batch == sequential."""
    from smrt_amd import make_model, make_snowpack, sensor_list
    from smrt_amd.runner.sequential_runner import SequentialRunner

    rng = np.random.default_rng(5)
    sps = [make_snowpack(np.append(rng.uniform(0.05, 0.3, 4), 100.0), "exponential", density=rng.uniform(150, 450, 5),
                         temperature=rng.uniform(230, 270, 5), corr_length=rng.uniform(5e-5, 3e-4, 5))
           for _ in range(4)]
    m = make_model("iba", "dort", rtsolver_options=dict(n_max_stream=16))
    res = m.run(sensor_list.amsre(), sps)
    assert res.data.dims == ("frequency", "snowpack", "polarization", "theta")
    assert res.data.shape == (6, 4, 2, 1)
    assert res.Tb(channel="37V").shape == (4,)
    seq = m.run(sensor_list.amsre(), sps, runner=SequentialRunner())
    assert np.array_equal(seq.data.values, res.data.values)
    df = res.to_dataframe()
    assert df.shape == (4, 12) and "89H" in df.columns
    one = m.run(sensor_list.amsre("19"), sps[2])
    assert abs(one.TbV() - res.Tb(channel="19V")[2]) == 0.0


def test_dmrt_model_run():
    from smrt_amd import make_model, make_snowpack, sensor_list

    d = load_golden("dmrt_L8_n16")
    sp = make_snowpack(d["thickness"], "sticky_hard_spheres", density=d["density"], temperature=d["temperature"],
                       radius=d["radius"], stickiness=d["stickiness"])
    m = make_model("dmrt_qca_shortrange", "dort", rtsolver_options=dict(n_max_stream=16))
    res = m.run(sensor_list.passive(list(d["frequency"]), list(d["theta_deg"])), sp)
    assert res.data.dims == ("frequency", "polarization", "theta")
    assert np.abs(res.data.values - d["result"]).max() < 1e-6


def test_error_handling_exception_and_nan():
    """Albedo >= 1 layer (smrt/test/test_dmrtdort.py:20-37 snowpack with dmrt_qca_shortrange): SMRTError by default,
    NaN result with error_handling='nan' (dort.py:327-334)."""
    from smrt_amd import SMRTError, make_model, make_snowpack, sensor_list

    sp = make_snowpack([0.1, 1000], "sticky_hard_spheres", density=[200, 400], temperature=[250.0, 250.0],
                       radius=[2e-4, 2e-4], stickiness=[0.1, 0.1])
    with pytest.raises(SMRTError, match="albedo"):
        make_model("dmrt_qca_shortrange", "dort").run(sensor_list.amsre("37V"), sp)
    res = make_model("dmrt_qca_shortrange", "dort", rtsolver_options=dict(error_handling="nan")).run(
        sensor_list.amsre("37V"), sp)
    assert np.isnan(res.TbV())


def test_emmodel_scalar_accessors():
    """IBA ks against the reference values and the MEMLS table (smrt/emmodel/test_iba.py:111-127)."""
    from smrt_amd import make_snowpack, sensor_list
    from smrt_amd.emmodel.iba import IBA

    d = load_golden("iba_ks_table")
    for row, memls in zip(d["table"], d["memls_reference"]):
        pc, ks, ka, er, ei, _ = row
        lay = make_snowpack([0.1], "exponential", density=300, temperature=265, corr_length=pc).layers[0]
        em = IBA(next(sensor_list.amsre("37V").iterate("frequency")) if False else sensor_list.passive(36.5e9, 55), lay)
        assert abs(em.ks(0).mean() - ks) < 1e-11 * ks and abs(em.ka - ka) < 1e-10 * ka
        assert abs(em.effective_permittivity() - (er + 1j * ei)) < 1e-13
        assert abs(em.ks(0).mean() - memls) < 0.01 * memls


def test_physics_nonscattering_limit_and_kirchhoff():
    """Physics invariants (smrt/test/test_physics_law.py, rtsolver/test_rtsolver.py:37-45 in spirit): an isothermal
    snowpack with vanishing scattering emits Tb = (1 - R_surface) T, so e_V >= e_H and Tb <= T; Rayleigh-Jeans and Planck agree."""
    from smrt_amd import make_model, make_snowpack, sensor_list

    T = 260.0
    sp = make_snowpack([1, 100], "exponential", density=[300, 300], temperature=T, corr_length=1e-7)
    m = make_model("iba", "dort", rtsolver_options=dict(rayleigh_jeans_approximation=True))
    res = m.run(sensor_list.passive(10e9, [20, 40, 55]), sp)
    tbv, tbh = res.TbV(), res.TbH()
    assert (tbv <= T + 1e-9).all() and (tbh <= tbv + 1e-9).all() and (tbv > 0.9 * T).all()
    # Rayleigh-Jeans versus Planck (smrt/rtsolver/test_rtsolver.py:122-136 in spirit): both agree with the oracle,
    # and differ from each other by far less than a kelvin at 10 GHz
    from oracle import dort_oracle as O

    spd = dict(thickness=np.array([1.0, 100.0]), density=np.array([300.0, 300.0]), temperature=np.array([T, T]),
               microstructure="exponential", corr_length=np.array([1e-7, 1e-7]))
    ref_rj = O.solve(spd, 10e9, [20, 40, 55], rayleigh_jeans=True)
    assert np.abs(res.data.values - ref_rj).max() < 1e-6
    res_pl = make_model("iba", "dort").run(sensor_list.passive(10e9, [20, 40, 55]), sp)
    assert np.abs(res_pl.data.values - O.solve(spd, 10e9, [20, 40, 55])).max() < 1e-6
    assert np.abs(res_pl.data.values - res.data.values).max() < 0.5


def _random_packs(rng, n, L=5):
    from smrt_amd import make_snowpack

    return [make_snowpack(np.append(rng.uniform(0.05, 0.3, L - 1), 100.0), "exponential",
                          density=rng.uniform(150, 450, L), temperature=rng.uniform(230, 270, L),
                          corr_length=rng.uniform(5e-5, 3e-4, L)) for _ in range(n)]


def test_zipped_sensors_run_only_the_listed_pairs():
    """A sequence of sensors zipped with a sequence of snowpacks (smrt/core/model.py:505-515): N pairs, not the N x N
    product -- the device gets a pair list (smrt_dort_run_pairs) -- and the same bits as the full-grid run."""
    from smrt_amd import make_model, sensor_list
    from smrt_amd._native import load_library

    rng = np.random.default_rng(9)
    sps = _random_packs(rng, 6)
    freqs = [10.65e9, 18.7e9, 36.5e9, 89e9, 18.7e9, 10.65e9]
    m = make_model("iba", "dort", rtsolver_options=dict(n_max_stream=16))
    grid = m.run(sensor_list.passive(sorted(set(freqs)), 55), sps)          # 4 frequencies x 6 snowpacks
    zipped = m.run([sensor_list.passive(f, 55) for f in freqs], sps)
    assert zipped.data.dims == ("snowpack", "polarization", "theta") and zipped.data.shape == (6, 2, 1)
    for k, f in enumerate(freqs):
        assert np.array_equal(zipped.data.values[k], grid.data.sel(frequency=f).values[k])
    # the generic runner protocol with an arbitrary list of pairs goes the same way
    from smrt_amd.runner.hip_batch_runner import HipBatchRunner

    pairs = [(sensor_list.passive(freqs[k], 55), sps[k]) for k in (4, 1, 1, 3)]
    res = HipBatchRunner()(m.run_single_simulation, [(p, None, "outer") for p in pairs])
    assert [float(r.TbV()) for r in res] == [float(zipped.TbV()[k]) for k in (4, 1, 1, 3)]
    assert load_library() is not None


def test_stacked_result_equals_nested_per_pair_results():
    """Model.run's fast path builds the (frequency, snowpack, ...) Result by reshaping the device output; the generic
    path nests one Result per pair with concat_results (smrt/core/result.py:768-817).  Same values, coordinates and
    diagnostics -- also for ragged layer counts (NaN below the last layer, like xr.concat)."""
    from smrt_amd import make_model, make_snowpack, sensor_list
    from smrt_amd.core.model import nest_results

    rng = np.random.default_rng(10)
    sps = _random_packs(rng, 3) + [make_snowpack([0.2, 50.0], "exponential", density=[250, 350], temperature=[255, 262],
                                                 corr_length=[1e-4, 2e-4])]
    m = make_model("iba", "dort", rtsolver_options=dict(n_max_stream=16))
    fast = m.run(sensor_list.amsre(), sps)
    plan = m.plan(sensor_list.amsre(), sps)
    slow = nest_results([m.run_single_simulation(pair, None, "none") for pair in plan.pairs()], plan.dimensions)
    assert fast.data.dims == slow.data.dims and np.array_equal(fast.data.values, slow.data.values)
    for k in slow.other_data:
        a, b = fast.other_data[k], slow.other_data[k]
        assert a.dims == b.dims, k
        np.testing.assert_array_equal(a.values, b.values, err_msg=k)
    assert np.isnan(fast.other_data["ks"].values[0, 3, 2:]).all() and np.isfinite(fast.other_data["ks"].values[0, 3, :2]).all()


def test_threaded_multi_device_path_and_growing_batches():
    """run_on_devices with one host thread per context (devices=[0, 0] exercises the threaded path on a one-GPU box)
    equals the single-context run bit for bit, with count-based and with cost-based shards, for a pair list too; a
    second, LARGER batch in the same process regrows every device buffer (the growth must happen on the
    context's own device)."""
    from smrt_amd._native import PackedBatch
    from smrt_amd.rtsolver.dort import run_on_devices

    rng = np.random.default_rng(21)
    for S in (40, 160):
        L = 6
        thick = np.concatenate([rng.uniform(0.05, 0.3, (S, L - 1)), np.full((S, 1), 100.0)], axis=1)
        b = PackedBatch([L] * S, thick, rng.uniform(150, 450, (S, L)) / 916.7, rng.uniform(230, 270, (S, L)),
                        rng.uniform(5e-5, 3e-4, (S, L)), None, [18.7e9, 36.5e9, 89e9], np.deg2rad([55.0]), n_max_stream=16)
        one = run_on_devices(b, devices=[0])
        two = run_on_devices(b, devices=[0, 0])
        cost = rng.uniform(1, 9, b.n_pairs)
        three = run_on_devices(b, devices=[0, 0, 0], cost=cost)
        for other in (two, three):
            assert np.array_equal(one.values, other.values) and np.array_equal(one.status, other.status)
            assert np.array_equal(one.layers, other.layers) and np.array_equal(one.streams, other.streams)
        pick = rng.permutation(b.n_pairs)[: b.n_pairs // 3]
        sub = run_on_devices(b, devices=[0, 0], pairs=pick)
        assert np.array_equal(sub.values, one.values[pick]) and np.array_equal(sub.layers, one.layers[pick])


def test_heterogeneous_snowpacks_through_the_model():
    """make_model([...one emmodel per layer...]) / a dict per medium / a layer's own emmodel, over a snowpack whose
    layers mix microstructure models (smrt/core/model.py:529-582), next to homogeneous snowpacks in the same run:
    the reference fixture, and batch == sequential."""
    from smrt_amd import make_model, make_snowpack, sensor_list
    from smrt_amd.runner.sequential_runner import SequentialRunner

    d = load_golden("mixed_L4_n16_passive")
    none = lambda a: [None if np.isnan(x) else float(x) for x in a]  # noqa: E731
    sp = make_snowpack(d["thickness"], [str(m) for m in d["microstructure"]], density=d["density"],
                       temperature=d["temperature"], corr_length=none(d["corr_length"]), radius=none(d["radius"]),
                       stickiness=none(d["stickiness"]))
    m = make_model([str(e) for e in d["emmodel"]], "dort", rtsolver_options=dict(n_max_stream=16))
    sensor = sensor_list.passive(list(d["frequency"]), list(d["theta_deg"]))
    res = m.run(sensor, [sp, sp])
    assert res.data.dims == ("frequency", "snowpack", "polarization", "theta")
    assert np.abs(res.data.values[:, 0] - d["result"]).max() < 1e-6 and np.array_equal(res.data.values[:, 0], res.data.values[:, 1])
    seq = m.run(sensor, [sp, sp], runner=SequentialRunner())
    assert np.array_equal(seq.data.values, res.data.values)
    # a layer's own emmodel: the same physics said differently
    sp2 = make_snowpack(d["thickness"], [str(mm) for mm in d["microstructure"]], density=d["density"],
                        temperature=d["temperature"], corr_length=none(d["corr_length"]), radius=none(d["radius"]),
                        stickiness=none(d["stickiness"]))
    sp2.layers[1].emmodel = "dmrt_qca_shortrange"
    res2 = make_model("iba", "dort", rtsolver_options=dict(n_max_stream=16)).run(sensor, sp2)
    assert np.array_equal(res2.data.values, res.data.values[:, 0])


def test_iba_dense_snow_correction_through_the_model():
    """emmodel_options=dict(dense_snow_correction="auto") (smrt/emmodel/iba.py:85-105): layers above half ice are solved
    on the inverted medium -- batch runner, sequential runner, the emmodel instance's own accessors -- against the
    reference fixtures (firn / ice-lens densities between ordinary snow layers; exponential passive, SHS active)."""
    from smrt_amd import make_model, make_snowpack, sensor_list
    from smrt_amd.emmodel.iba import IBA
    from smrt_amd.runner.sequential_runner import SequentialRunner

    d = load_golden("iba_dense_auto_L5_n12")
    sp = make_snowpack(d["thickness"], "exponential", density=d["density"], temperature=d["temperature"],
                       corr_length=d["corr_length"])
    opts = dict(emmodel_options=dict(dense_snow_correction="auto"), rtsolver_options=dict(n_max_stream=12))
    m = make_model("iba", "dort", **opts)
    sensor = sensor_list.passive(list(d["frequency"]), list(d["theta_deg"]))
    res = m.run(sensor, [sp, sp])
    assert np.abs(res.data.values[:, 0] - d["result"]).max() < 1e-6
    seq = m.run(sensor, sp, runner=SequentialRunner())
    assert np.array_equal(seq.data.values, res.data.values[:, 0])
    np.testing.assert_allclose(np.asarray(seq.other_data["ks"].values)[0].ravel(), d["f0_ks"], rtol=1e-11)
    # without the option the dense layers keep ice inclusions in air: a different answer (the option is not a no-op here)
    plain = make_model("iba", "dort", rtsolver_options=dict(n_max_stream=12)).run(sensor, sp)
    assert np.abs(plain.data.values - d["result"]).max() > 0.5
    e = IBA(sensor_list.passive(float(d["frequency"][1]), 55), sp.layers[3], dense_snow_correction="auto")
    assert abs(e.frac_volume - (1.0 - sp.layers[3].frac_volume)) < 1e-15      # iba.py:98-99: the inverted layer's
    np.testing.assert_allclose([e._ks, e.ka], [d["f1_ks"][3], d["f1_ka"][3]], rtol=1e-10)
    da = load_golden("iba_dense_auto_shs_active_L3_n8")
    spa = make_snowpack(da["thickness"], "sticky_hard_spheres", density=da["density"], temperature=da["temperature"],
                        radius=da["radius"], stickiness=da["stickiness"])
    ma = make_model("iba", "dort", emmodel_options=dict(dense_snow_correction="auto"), rtsolver_options=dict(n_max_stream=8, m_max=2))
    ra = ma.run(sensor_list.active(float(da["frequency"][0]), list(da["theta_inc_deg"])), spa)
    assert_backscatter_close(ra.data.values, da["result"][0], spread=reference_method_spread(da)[0])


def test_emmodel_ft_even_phase_on_the_device():
    """The last piece of the emmodel protocol (smrt/rtsolver/dort.py:231-247 consumes it): ft_even_phase(mu_s, mu_i,
    m_max, npol) of smrt_amd's emmodel classes, evaluated by the device, against the oracle -- IBA on both
    microstructure models and a Rayleigh emmodel, both hemispheres, active and passive shapes."""
    from oracle import dort_oracle as O
    from smrt_amd import make_snowpack, sensor_list
    from smrt_amd.core.error import SMRTError
    from smrt_amd.emmodel.dmrt_qca_shortrange import DMRT_QCA_ShortRange
    from smrt_amd.emmodel.iba import IBA

    mu = np.array([0.97, 0.7, 0.35, 0.1])
    mu_full = np.concatenate([mu, -mu])
    cases = [(IBA, "iba", make_snowpack([1.0], "exponential", density=[300], temperature=[262], corr_length=[2e-4]),
              dict(microstructure="exponential", corr_length=[2e-4])),
             (IBA, "iba", make_snowpack([1.0], "sticky_hard_spheres", density=[300], temperature=[262], radius=[1.5e-4], stickiness=[0.25]),
              dict(microstructure="sticky_hard_spheres", radius=[1.5e-4], stickiness=[0.25])),
             (DMRT_QCA_ShortRange, "dmrt_qca_shortrange",
              make_snowpack([1.0], "sticky_hard_spheres", density=[300], temperature=[262], radius=[1.5e-4], stickiness=[0.25]),
              dict(microstructure="sticky_hard_spheres", radius=[1.5e-4], stickiness=[0.25]))]
    # ... and IBA on the inverted medium (dense_snow_correction="auto" above half ice), both microstructure models
    dense = lambda **kw: (lambda sensor, layer: IBA(sensor, layer, dense_snow_correction="auto"))   # noqa: E731
    cases += [(dense(), "iba_dense_auto", make_snowpack([1.0], "exponential", density=[700], temperature=[262], corr_length=[2e-4]),
               dict(microstructure="exponential", corr_length=[2e-4])),
              (dense(), "iba_dense_auto",
               make_snowpack([1.0], "sticky_hard_spheres", density=[700], temperature=[262], radius=[1.5e-4], stickiness=[0.25]),
               dict(microstructure="sticky_hard_spheres", radius=[1.5e-4], stickiness=[0.25]))]
    for cls, name, sp, osp in cases:
        layer = O.make_layers(name, 36.5e9, dict(thickness=[1.0], density=[float(sp.layers[0].density)], temperature=[262.0], **osp))[0]
        em_a = cls(sensor_list.active(36.5e9, 40), sp.layers[0])
        got = em_a.ft_even_phase(mu_full, mu_full, 2)
        assert got.shape == (3, 3, 3, 8, 8)
        ref = np.asarray(layer.ft_even_phase(mu_full, mu_full, 2, 3))
        np.testing.assert_allclose(got, ref, rtol=1e-10, atol=1e-13 * np.abs(ref).max())   # sums that cancel: absolute floor
        em_p = cls(sensor_list.passive(36.5e9, 55), sp.layers[0])
        got = em_p.ft_even_phase(mu, mu_full, 0)
        assert got.shape == (2, 2, 1, 4, 8)
        np.testing.assert_allclose(got, np.asarray(layer.ft_even_phase(mu, mu_full, 0, 2)), rtol=1e-10)
        # energy conservation of mode 0 (what the reference's test_iba.py checks): the (V, H) column sums integrate to ks
        assert np.isclose(em_p.ks(mu)[0, 0], layer.ks, rtol=1e-11)
    with pytest.raises(SMRTError):
        em_a.ft_even_phase(mu, np.array([1.0, 0.5]), 2)     # mu_i = 1 with three polarisations (common.py:389-390)


def test_emmodels_without_a_device_implementation_through_the_model():
    """make_model("rayleigh" | "prescribed_kskaeps", "dort").run(...): emmodels evaluated on the host (smrt_amd's
    rtsolver asks them for effective_permittivity / ks / ka / ft_even_phase like smrt/rtsolver/dort.py does) feeding the
    device solver, against the reference; then the same route with a user-defined emmodel class wrapped around IBA,
    which must agree with the device's own IBA."""
    from conftest import COHERENT_HOST_FIXTURES, HOST_EMMODEL_FIXTURES, fixture_coherent, model_snowpack_from_fixture
    from smrt_amd import make_model, make_snowpack, sensor_list
    from smrt_amd.emmodel.iba import IBA

    for name in HOST_EMMODEL_FIXTURES + COHERENT_HOST_FIXTURES:   # (the latter: together with process_coherent_layers)
        d = load_golden(name)
        sp = model_snowpack_from_fixture(d)
        act = str(d["mode"]) == "A"
        opts = dict(n_max_stream=int(d["opt_n_max_stream"]), process_coherent_layers=fixture_coherent(d))
        if act:
            opts["m_max"] = int(d["opt_m_max"])
            sensor = sensor_list.active(list(d["frequency"]), list(d["theta_inc_deg"]))
        else:
            sensor = sensor_list.passive(list(d["frequency"]), list(d["theta_deg"]))
        res = make_model(str(d["emmodel"]), "dort", rtsolver_options=opts).run(sensor, sp)
        for i, f in enumerate(d["frequency"]):
            sel = dict(frequency=f) if len(d["frequency"]) > 1 else {}   # a single frequency is not a dimension
            if act:
                got = np.array([res.sigmaVV(**sel), res.sigmaHH(**sel)])
                ref = 4 * np.pi * np.cos(np.deg2rad(d["theta_inc_deg"])) * np.array([d["result"][i, 0, 0], d["result"][i, 1, 1]])
                np.testing.assert_allclose(np.squeeze(got), ref, rtol=1e-8)
            else:
                np.testing.assert_allclose(np.ravel(res.TbV(**sel)), d["result"][i, 0], atol=1e-6)
                np.testing.assert_allclose(np.ravel(res.TbH(**sel)), d["result"][i, 1], atol=1e-6)
        ks_last = np.ravel(res.other_data["ks"].values)[-len(d["thickness"]):]     # (NaN after the layers that stayed)
        ks_ref = d["f%d_ks" % (len(d["frequency"]) - 1)]
        np.testing.assert_allclose(ks_last[:len(ks_ref)], ks_ref, rtol=1e-12)

    class WrappedIBA:   # no device_name: goes through the host route
        def __init__(self, sensor, layer):
            self.inner = IBA(sensor, layer)
            self.ka = self.inner.ka

        def effective_permittivity(self):
            return self.inner.effective_permittivity()

        def ks(self, mu, npol=2):
            return self.inner.ks(mu, npol)

        def ft_even_phase(self, mu_s, mu_i, m_max, npol=None):
            return self.inner.ft_even_phase(mu_s, mu_i, m_max, npol=npol)

    sp = make_snowpack([0.2, 0.5, 100], "exponential", density=[220, 300, 380], temperature=[255, 260, 268],
                       corr_length=[8e-5, 2e-4, 1.5e-4])
    sensor = sensor_list.passive([18.7e9, 36.5e9], 55)
    opts = dict(n_max_stream=16)
    native = make_model("iba", "dort", rtsolver_options=opts).run(sensor, sp)
    wrapped = make_model(WrappedIBA, "dort", rtsolver_options=opts).run(sensor, sp)
    np.testing.assert_allclose(wrapped.TbV(), native.TbV(), atol=1e-7)
    np.testing.assert_allclose(wrapped.TbH(), native.TbH(), atol=1e-7)
    # a layer list that mixes a device emmodel with a host one takes the host route as a whole
    mixed = make_model([WrappedIBA, "iba", WrappedIBA], "dort", rtsolver_options=opts).run(sensor, sp)
    np.testing.assert_allclose(mixed.TbV(), native.TbV(), atol=1e-7)
    # ... also when the device emmodel of that list carries an option that changes its device name: IBA under
    # dense_snow_correction="auto" on a layer above half ice ("iba_inverted" on the device) keeps the option on the host
    # route (it was instantiated as a plugin named 'iba_inverted', without its options)
    dense = make_snowpack([0.2, 0.5, 100], "exponential", density=[220, 700, 380], temperature=[255, 260, 268],
                          corr_length=[8e-5, 2e-4, 1.5e-4])
    auto = dict(dense_snow_correction="auto")
    native = make_model("iba", "dort", rtsolver_options=opts, emmodel_options=auto).run(sensor, dense)
    mixed = make_model([WrappedIBA, "iba", WrappedIBA], "dort", rtsolver_options=opts,
                       emmodel_options=[{}, auto, {}]).run(sensor, dense)
    np.testing.assert_allclose(mixed.TbV(), native.TbV(), atol=1e-7)
    np.testing.assert_allclose(mixed.TbH(), native.TbH(), atol=1e-7)
    plain = make_model("iba", "dort", rtsolver_options=opts).run(sensor, dense)
    assert np.abs(np.asarray(plain.TbV()) - np.asarray(native.TbV())).max() > 1e-3    # (the option does matter here)


def test_process_coherent_layers_through_the_model():
    """rtsolver_options=dict(process_coherent_layers=True) through make_model()/Model.run() -- with ALL the frequencies
    of the fixture in one sensor, which the reference itself cannot run (its test on the last layer breaks on arrays,
    smrt/interface/coherent_flat.py:26): every frequency against the reference's single-frequency runs, and the
    diagnostics of each frequency list the layers that frequency kept."""
    from conftest import COHERENT_FIXTURES, model_snowpack_from_fixture
    from smrt_amd import make_model, sensor_list

    d = load_golden(COHERENT_FIXTURES[0])
    sp = model_snowpack_from_fixture(d)
    m = make_model("iba", "dort", rtsolver_options=dict(n_max_stream=int(d["opt_n_max_stream"]), process_coherent_layers=True))
    res = m.run(sensor_list.passive(list(d["frequency"]), list(d["theta_deg"])), sp)
    for i, f in enumerate(d["frequency"]):
        np.testing.assert_allclose(np.ravel(res.TbV(frequency=f)), d["result"][i, 0], atol=1e-6)
        np.testing.assert_allclose(np.ravel(res.TbH(frequency=f)), d["result"][i, 1], atol=1e-6)
    ks = np.asarray(res.other_data["ks"].values)           # (frequency, layer), NaN after the kept layers
    th = np.asarray(res.other_data["thickness"].values)
    for i in range(len(d["frequency"])):
        kept = len(d["f%d_ks" % i])
        np.testing.assert_allclose(ks[i, :kept], d["f%d_ks" % i], rtol=1e-10)
        assert np.isnan(ks[i, kept:]).all() and np.isnan(th[i, kept:]).all()
        assert 0.002 not in th[i, :kept]                    # the 2 mm crust left at every frequency,
        assert (0.003 in th[i, :kept]) == (i == 2)          # the 3 mm ice lens only below 36.5 GHz
    # one simulation at a time (the rtsolver protocol) gives the same
    one = m.run(sensor_list.passive(float(d["frequency"][2]), list(d["theta_deg"])), sp)
    np.testing.assert_allclose(np.ravel(one.TbV()), d["result"][2, 0], atol=1e-6)
    assert len(np.ravel(one.other_data["ks"].values)) == len(d["f2_ks"])


def test_rough_substrate_through_the_model():
    """A substrate object without a device implementation -- here one that answers with what the reference's
    geometrical_optics / iem_fung92 substrates answered (stored in the fixtures) -- under the snowpack of an active
    simulation: smrt_amd's rtsolver asks it for its specular and diffuse reflection on the streams of the last layer (which
    it gets from a cheap pre-pass of the device emmodels), builds the dense bottom reflection of every azimuth mode like
    compute_interface_properties does and the device starts its recursion from it.  Against the reference's backscatter."""
    from conftest import ROUGH_SUBSTRATE_FIXTURES, model_snowpack_from_fixture
    from smrt_amd import make_model, sensor_list
    from smrt_amd.core.snowpack import Snowpack

    for name in ROUGH_SUBSTRATE_FIXTURES:
        d = load_golden(name)

        class FromFixture:
            temperature = 268.0

            def specular_reflection_matrix(self, frequency, eps_1, mu1, npol):
                np.testing.assert_allclose(mu1, d["sub_mu"], rtol=1e-11)      # the streams the reference had
                return d["sub_spec_raw"]

            def ft_even_diffuse_reflection_matrix(self, frequency, eps_1, mu_s, mu_i, m_max, npol):
                return d["sub_diff_raw"]

        sp = model_snowpack_from_fixture(d)
        sp = Snowpack(layers=sp.layers, substrate=FromFixture())
        opts = dict(n_max_stream=int(d["opt_n_max_stream"]), m_max=int(d["opt_m_max"]))
        res = make_model("iba", "dort", rtsolver_options=opts).run(
            sensor_list.active(float(d["frequency"][0]), list(d["theta_inc_deg"])), sp)
        fac = 4 * np.pi * np.cos(np.deg2rad(d["theta_inc_deg"]))
        np.testing.assert_allclose(np.ravel(res.sigmaVV()), fac * d["result"][0, 0, 0], rtol=1e-8)
        np.testing.assert_allclose(np.ravel(res.sigmaHH()), fac * d["result"][0, 1, 1], rtol=1e-8)


def test_rough_interfaces_through_the_model():
    """Interface objects without a device implementation -- here ones that answer with what the reference's iem_fung92 /
    geometrical_optics interfaces answered (stored in the fixtures) -- at the surface or between two layers, through
    make_snowpack(interface=[...]) and Model.run: the rtsolver places the streams (pre-pass of the device emmodels), asks the
    object for its specular / coherent and diffuse matrices, combines them like compute_interface_properties and the
    device composes the dense interface with the layers below.  Against the reference, passive and active."""
    from conftest import ROUGH_INTERFACE_FIXTURES, ReplayInterface, snowpack_dict
    from smrt_amd import make_model, make_snowpack, sensor_list
    from smrt_amd.interface.flat import Flat

    for name in ROUGH_INTERFACE_FIXTURES:
        d = load_golden(name)
        sp = snowpack_dict(d)
        i = int(d["rough_interface"][0])
        itf = [Flat()] * len(sp["thickness"])
        itf[i] = ReplayInterface(d)
        pack = make_snowpack(sp["thickness"], "exponential", density=sp["density"], temperature=sp["temperature"],
                             corr_length=sp["corr_length"], interface=itf)
        f = float(d["frequency"][0])
        if str(d["mode"]) == "A":
            opts = dict(n_max_stream=int(d["opt_n_max_stream"]), m_max=int(d["opt_m_max"]))
            res = make_model("iba", "dort", rtsolver_options=opts).run(sensor_list.active(f, list(d["theta_inc_deg"])), pack)
            fac = 4 * np.pi * np.cos(np.deg2rad(d["theta_inc_deg"]))
            np.testing.assert_allclose(np.ravel(res.sigmaVV()), fac * d["result"][0, 0, 0], rtol=1e-8)
            np.testing.assert_allclose(np.ravel(res.sigmaHH()), fac * d["result"][0, 1, 1], rtol=1e-8)
        else:
            opts = dict(n_max_stream=int(d["opt_n_max_stream"]))
            res = make_model("iba", "dort", rtsolver_options=opts).run(sensor_list.passive(f, list(d["theta_deg"])), pack)
            np.testing.assert_allclose(np.ravel(res.TbV()), d["result"][0, 0], atol=1e-6)
            np.testing.assert_allclose(np.ravel(res.TbH()), d["result"][0, 1], atol=1e-6)


def test_rough_substrate_passive_through_the_model():
    """The same host-evaluated substrate route in passive mode (two polarisations, mode 0, emissivity_matrix for the
    emission of the substrate), on the rough substrates the reference runs there."""
    from conftest import ROUGH_SUBSTRATE_PASSIVE_FIXTURES, model_snowpack_from_fixture
    from smrt_amd import make_model, sensor_list
    from smrt_amd.core.snowpack import Snowpack

    for name in ROUGH_SUBSTRATE_PASSIVE_FIXTURES:
        d = load_golden(name)

        class FromFixture:
            temperature = float(d["substrate_temperature"])

            def specular_reflection_matrix(self, frequency, eps_1, mu1, npol):
                np.testing.assert_allclose(mu1, d["sub_mu"], rtol=1e-11)
                return d["sub_spec_raw"]

            def ft_even_diffuse_reflection_matrix(self, frequency, eps_1, mu_s, mu_i, m_max, npol):
                return d["sub_diff_raw"]

            def emissivity_matrix(self, frequency, eps_1, mu1, npol):
                return d["sub_emis_raw"]

        sp = Snowpack(layers=model_snowpack_from_fixture(d).layers, substrate=FromFixture())
        res = make_model("iba", "dort", rtsolver_options=dict(n_max_stream=int(d["opt_n_max_stream"]))).run(
            sensor_list.passive(float(d["frequency"][0]), list(d["theta_deg"])), sp)
        np.testing.assert_allclose(np.ravel(res.TbV()), d["result"][0, 0], atol=1e-6)
        np.testing.assert_allclose(np.ravel(res.TbH()), d["result"][0, 1], atol=1e-6)


@pytest.mark.parametrize("name", ["headline_shape_two_snowpacks", "flat_substrate_and_atmosphere", "dmrt_on_a_reflector",
                                  "active_dense_auto"])
def test_reference_shaped_objects_through_the_runner(name):
    """What the reference's `Model.run(..., runner=HipBatchRunner())` hands over (smrt/core/model.py:395-398), replayed
    on the GPU box -- which has no smrt package -- as stand-ins that carry the class identities and public attributes
    dumped from REAL smrt objects (tests/golden/reference_objects.json): bound `run_single_simulation` of an smrt-shaped
    Model, ((sensor, snowpack), atmosphere, parallel_computation) triples.  One Result per item, in order, equal to
    what smrt's own iba | dmrt + dort returned for those objects; the rtsolver protocol (DORT.solve per simulation with
    stand-in emmodel instances) gives the same bits."""
    from conftest import load_reference_objects, standins_from_dump
    from smrt_amd.runner.hip_batch_runner import HipBatchRunner

    case = load_reference_objects()[name]
    model, sims, packs, want = standins_from_dump(case)
    results = HipBatchRunner()(model.run_single_simulation, [(sim, None, "outer") for sim in sims])
    assert len(results) == len(sims)
    for (sensor, sp), res, ref in zip(sims, results, want):
        got = np.asarray(res.data.values)
        assert got.shape == ref.shape
        if sensor.mode == "P":
            assert np.abs(got - ref).max() < 1e-6
        else:
            assert_backscatter_close(got, ref)
        np.testing.assert_allclose(res.other_data["thickness"].values, [lay.thickness for lay in sp.layers])
    # the rtsolver protocol: what Model.run_single_simulation does per item (model.py:596-617)
    from smrt_amd.rtsolver.dort import DORT

    sensor, sp = sims[-1]
    emmodels = [model.emmodel(sensor, layer, **model.emmodel_options) for layer in sp.layers]
    if name == "active_dense_auto":   # the reference's IBA instance keeps the volume fraction it worked with (iba.py:96-99)
        for em, layer in zip(emmodels, sp.layers):
            em.frac_volume = 1.0 - layer.frac_volume if layer.frac_volume > 0.5 else layer.frac_volume
    one = DORT(**model.rtsolver_options).solve(sp, emmodels, sensor, sp.atmosphere)
    assert np.array_equal(np.asarray(one.data.values), np.asarray(results[-1].data.values))


def test_own_rough_models_end_to_end():
    """Rough interfaces and rough substrates by NAME, standalone: make_interface("iem_fung92" | "geometrical_optics", ...)
    in make_snowpack(interface=[...]) and make_soil("iem_fung92" | "geometrical_optics" | "geometrical_optics_backscatter",
    eps, T, ...) under the snowpack -- smrt_amd's own evaluators (interface/iem_fung92.py, geometrical_optics*.py) feeding
    the dense interface / bottom-boundary composition of the device -- against the reference's results for the same
    models and parameters (the fixtures of the replayed-object tests above), passive and active."""
    import warnings

    from conftest import ROUGH_INTERFACE_MODELS, ROUGH_SUBSTRATE_MODELS, snowpack_dict
    from smrt_amd import make_interface, make_model, make_snowpack, make_soil, sensor_list

    def check(d, pack):
        f = float(d["frequency"][0])
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            if str(d["mode"]) == "A":
                opts = dict(n_max_stream=int(d["opt_n_max_stream"]), m_max=int(d["opt_m_max"]))
                res = make_model("iba", "dort", rtsolver_options=opts).run(sensor_list.active(f, list(d["theta_inc_deg"])), pack)
                fac = 4 * np.pi * np.cos(np.deg2rad(d["theta_inc_deg"]))
                np.testing.assert_allclose(np.ravel(res.sigmaVV()), fac * d["result"][0, 0, 0], rtol=1e-8)
                np.testing.assert_allclose(np.ravel(res.sigmaHH()), fac * d["result"][0, 1, 1], rtol=1e-8)
            else:
                opts = dict(n_max_stream=int(d["opt_n_max_stream"]))
                res = make_model("iba", "dort", rtsolver_options=opts).run(sensor_list.passive(f, list(d["theta_deg"])), pack)
                np.testing.assert_allclose(np.ravel(res.TbV()), d["result"][0, 0], atol=1e-6)
                np.testing.assert_allclose(np.ravel(res.TbH()), d["result"][0, 1], atol=1e-6)

    for name, (model, kw) in ROUGH_INTERFACE_MODELS.items():
        d = load_golden(name)
        sp = snowpack_dict(d)
        itf = ["flat"] * len(sp["thickness"])
        itf[int(d["rough_interface"][0])] = make_interface(model, **kw)
        check(d, make_snowpack(sp["thickness"], "exponential", density=sp["density"], temperature=sp["temperature"],
                               corr_length=sp["corr_length"], interface=itf))
    for name, (model, kw) in ROUGH_SUBSTRATE_MODELS.items():
        d = load_golden(name)
        sp = snowpack_dict(d)
        check(d, make_snowpack(sp["thickness"], "exponential", density=sp["density"], temperature=sp["temperature"],
                               corr_length=sp["corr_length"], substrate=make_soil(model, complex(8.0, 1.0), 268.0, **kw)))


@pytest.mark.parametrize("name", ["rough_prune_iem_L4_n10_passive", "rough_prune_go_L4_n10_active",
                                  "rough_coherent_iem_L5_n10_passive", "rough_coherent_adjacent_L5_n10_passive",
                                  "rough_coherent_gosub_L4_n10_active"])
def test_rough_interfaces_under_prune_and_coherent_options(name):
    """The two parity holes of round 3's rough-interface work, closed: prune_deep_snowpack cutting above a rough interface
    (the reference's truncated system keeps that interface's dense reflection, smrt/rtsolver/dort.py:443-452 with
    rtsolver_utils.py:567-597) and process_coherent_layers together with rough interfaces / a rough substrate (sampled on
    the streams of the reduced snowpack; an interface on the collapsed layer goes into the reference's CoherentFlat,
    interface/coherent_flat.py:16-57).  make_snowpack / make_interface / make_soil by name -> Model.run -> the GPU,
    against the reference's own results (tests/golden/make_rough_option_fixtures.py)."""
    from conftest import ROUGH_OPTION_CASES
    from test_hostemu_kernel import check_rough_option_case

    check_rough_option_case(name, ROUGH_OPTION_CASES[name])


def test_wet_snow_through_the_model():
    """Wet snow standalone: make_snowpack(volumetric_liquid_water=... | liquid_water=...) -> Model.run on the device emmodels
    (ice grains coated in water: Maxwell Garnett in a water host, smrt/permittivity/wetice.py:12-45, water.py:14-43),
    against the reference's results -- IBA passive, IBA active with a very wet layer inverted by dense_snow_correction=
    "auto", DMRT-QCA passive -- and the layer diagnostics (effective permittivity, ks, ka) of the wet layers."""
    from conftest import WET_FIXTURES, reference_method_spread
    from smrt_amd import make_model, make_snowpack, sensor_list

    for name in WET_FIXTURES:
        d = load_golden(name)
        ms = str(d["microstructure"])
        kw = dict(corr_length=d["corr_length"]) if ms == "exponential" else dict(radius=d["radius"], stickiness=d["stickiness"])
        sp = make_snowpack(d["thickness"], ms, density=d["density"], temperature=d["temperature"],
                           liquid_water=list(d["liquid_water"]), **kw)
        np.testing.assert_allclose([lay.frac_volume for lay in sp.layers], d["frac_volume"], rtol=1e-13)
        em = str(d["emmodel"])
        em_opts = dict(dense_snow_correction="auto") if em == "iba_dense_auto" else None
        opts = dict(n_max_stream=int(d["opt_n_max_stream"]))
        if str(d["mode"]) == "A":
            opts["m_max"] = int(d["opt_m_max"])
            sensor = sensor_list.active(list(d["frequency"]), list(d["theta_inc_deg"]))
        else:
            sensor = sensor_list.passive(list(d["frequency"]), list(d["theta_deg"]))
        res = make_model("iba" if em.startswith("iba") else em, "dort", rtsolver_options=opts, emmodel_options=em_opts).run(sensor, sp)
        got = np.asarray(res.data.values).reshape(d["result"].shape)
        if str(d["mode"]) == "A":
            assert_backscatter_close(got, d["result"], spread=reference_method_spread(d))
        else:
            assert np.abs(got - d["result"]).max() < 1e-6
        last = len(d["frequency"]) - 1
        ks = np.asarray(res.other_data["ks"].values).reshape(len(d["frequency"]), -1)[last]
        eps = np.asarray(res.other_data["effective_permittivity"].values).reshape(len(d["frequency"]), -1)[last]
        np.testing.assert_allclose(ks, d["f%d_ks" % last], rtol=1e-10)
        np.testing.assert_allclose(eps, d["f%d_effective_permittivity" % last], rtol=1e-11)


def test_other_microstructure_models_through_the_model():
    """IBA over teubner_strey and independent_sphere layers (device: ft_corr, dort_physics.hpp) mixed with exponential and
    sticky-hard-spheres ones, and over the three models on the unified parameters (reparametrisations of the same closed
    forms, core/layer.py), one microstructure model per layer through make_snowpack's list form, against the reference,
    passive and active; a uniform teubner_strey snowpack takes the uniform-microstructure batch (no per-layer codes)."""
    from conftest import MICRO_FIXTURES, reference_method_spread
    from smrt_amd import make_model, make_snowpack, sensor_list

    none = lambda a: [None if np.isnan(x) else float(x) for x in a]   # noqa: E731
    for name in MICRO_FIXTURES:
        d = load_golden(name)
        micro_args = {k: none(d[k]) for k in ("corr_length", "radius", "stickiness", "repeat_distance", "porod_length",
                                               "polydispersity") if k in d}   # (the unified fixtures: porod length, polydispersity)
        sp = make_snowpack(d["thickness"], [str(m) for m in d["microstructure"]], density=d["density"], temperature=d["temperature"],
                           **micro_args)
        opts = dict(n_max_stream=int(d["opt_n_max_stream"]))
        if str(d["mode"]) == "A":
            opts["m_max"] = int(d["opt_m_max"])
            res = make_model("iba", "dort", rtsolver_options=opts).run(sensor_list.active(list(d["frequency"]), list(d["theta_inc_deg"])), sp)
            assert_backscatter_close(np.asarray(res.data.values).reshape(d["result"].shape), d["result"], spread=reference_method_spread(d))
        else:
            res = make_model("iba", "dort", rtsolver_options=opts).run(sensor_list.passive(list(d["frequency"]), list(d["theta_deg"])), sp)
            assert np.abs(np.asarray(res.data.values).reshape(d["result"].shape) - d["result"]).max() < 1e-6
        last = len(d["frequency"]) - 1
        np.testing.assert_allclose(np.asarray(res.other_data["ks"].values).reshape(len(d["frequency"]), -1)[last], d["f%d_ks" % last], rtol=1e-10)
    from oracle import dort_oracle as O

    ts = make_snowpack([0.3, 100.0], "teubner_strey", density=[300, 380], temperature=[260, 265], corr_length=[1.5e-4, 2e-4],
                       repeat_distance=[1e-3, 1.5e-3])
    res = make_model("iba", "dort", rtsolver_options=dict(n_max_stream=16)).run(sensor_list.passive(36.5e9, 55), ts)
    ref = O.solve(dict(thickness=np.array([0.3, 100.0]), density=np.array([300.0, 380.0]), temperature=np.array([260.0, 265.0]),
                       microstructure="teubner_strey", corr_length=np.array([1.5e-4, 2e-4]), repeat_distance=np.array([1e-3, 1.5e-3])),
                  36.5e9, [55.0], n_max_stream=16)
    assert np.abs(np.ravel(res.data.values) - np.ravel(ref)).max() < 1e-6


@pytest.mark.gpu
def test_an_emmodel_of_the_iba_family_hands_over_its_scalars():
    """SMRT_EM_IBA_HOST through Model.run: an emmodel class whose phase matrix is IBA's and whose scalars are its own --
    here IBA_original's and IBA_MaxwellGarnett's numbers, evaluated by a class of this test file from the oracle's layer
    objects (the GPU box has no reference package; with the reference's own classes: tests/test_reference_binding.py) --
    declares `iba_phase_family`, and the solver takes effective permittivity, ks, ka and the phase coefficient from the
    object and assembles the phase matrices on the device: the reference's results for these emmodels (fixtures)."""
    from conftest import IBA_FAMILY_FIXTURES, reference_method_spread
    from oracle import dort_oracle as O
    from smrt_amd import make_model, make_snowpack, sensor_list

    def family_member(oracle_class, family=True):
        class Member:
            iba_phase_family = family        # phase matrix = iba_coeff x FT of the autocorrelation function x Rayleigh geometry
                                             # ("complex_k": at the complex wavenumber of the strong-contrast expansion)

            def __init__(self, sensor, layer):
                ms = layer.microstructure
                params = {k: getattr(ms, k) for k in ("corr_length", "radius", "stickiness", "porod_length", "polydispersity") if hasattr(ms, k)}
                em = oracle_class(float(sensor.frequency), layer.frac_volume, layer.temperature, layer.microstructure_model, **params)
                self.frac_volume, self.microstructure = layer.frac_volume, ms
                self.iba_coeff, self.ka, self._ks, self._eps = em.iba_coeff, em.ka, em.ks, em.eps_eff

            def effective_permittivity(self):
                return self._eps

            def ks(self, mu, npol=2):
                return np.full((npol, np.size(mu)), self._ks)
        return Member

    members = {"iba_original": family_member(O.IBAOriginalLayer), "iba_maxwell_garnett": family_member(O.IBAMaxwellGarnettLayer),
               "symsce_torquato21": family_member(O.SymSCELayer, "complex_k")}
    for name in IBA_FAMILY_FIXTURES:
        d = load_golden(name)
        none = lambda a: [None if np.isnan(x) else float(x) for x in a]   # noqa: E731
        ms = str(d["microstructure"]) if np.ndim(d["microstructure"]) == 0 else [str(m) for m in d["microstructure"]]
        args = {k: none(d[k]) for k in ("corr_length", "radius", "stickiness", "repeat_distance", "porod_length", "polydispersity")
                if k in d}
        sp = make_snowpack(d["thickness"], ms, density=d["density"], temperature=d["temperature"], **args)
        em = [members.get(str(e), str(e)) for e in np.atleast_1d(d["emmodel"])]
        em = em[0] if len(em) == 1 else em
        opts = dict(n_max_stream=int(d["opt_n_max_stream"]))
        if str(d["mode"]) == "A":
            opts["m_max"] = int(d["opt_m_max"])
            res = make_model(em, "dort", rtsolver_options=opts).run(sensor_list.active(list(d["frequency"]), list(d["theta_inc_deg"])), sp)
            assert_backscatter_close(np.asarray(res.data.values).reshape(d["result"].shape), d["result"], spread=reference_method_spread(d))
        else:
            res = make_model(em, "dort", rtsolver_options=opts).run(sensor_list.passive(list(d["frequency"]), list(d["theta_deg"])), sp)
            assert np.abs(np.asarray(res.data.values).reshape(d["result"].shape) - d["result"]).max() < 1e-6
