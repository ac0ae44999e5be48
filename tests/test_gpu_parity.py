"""Parity of the HIP path (through the C ABI) with the reference golden vectors and the CPU oracle.  GPU only."""
import numpy as np
import pytest

from conftest import PASSIVE_FIXTURES, fixture_options, load_golden, snowpack_dict

pytestmark = pytest.mark.gpu

TB_TOL = 1e-6  # K, BASELINE.json north_star


@pytest.fixture(scope="module")
def ctx():
    from smrt_amd._native import DortContext

    c = DortContext(0)
    yield c
    c.close()


def batch_from_fixture(d, freqs=None):
    from smrt_amd._native import PackedBatch

    sp = snowpack_dict(d)
    ms = sp["microstructure"]
    p1 = sp["corr_length"] if ms == "exponential" else sp["radius"]
    p2 = None if ms == "exponential" else np.broadcast_to(sp["stickiness"], p1.shape)
    fr = d["frequency"] if freqs is None else d["frequency"][freqs]
    o = fixture_options(d)
    return PackedBatch([len(sp["thickness"])], sp["thickness"], sp["frac_volume"], sp["temperature"], p1, p2, fr,
                       np.deg2rad(d["theta_deg"]), emmodel=str(d["emmodel"]), microstructure=ms,
                       n_max_stream=o["n_max_stream"])


@pytest.mark.parametrize("name", PASSIVE_FIXTURES)
@pytest.mark.parametrize("threads", [64, 256, 512])
def test_passive_golden(ctx, name, threads):
    d = load_golden(name)
    if "n64" in name and threads != 512:
        pytest.skip("the 64-stream (global-workspace) kernel has a fixed workgroup size")
    ctx.set_block_threads(threads)
    out = ctx.run(batch_from_fixture(d))
    ctx.set_block_threads(0)
    assert (out.status == 0).all(), out.status
    assert np.abs(out.values - d["result"]).max() < TB_TOL
    # Result.other_data counterparts (rtsolver_utils.py:338-342,373-398)
    L = len(d["thickness"])
    for i in range(len(d["frequency"])):
        tag = "f%d_" % i
        eps = out.layers[i, :L, 0] + 1j * out.layers[i, :L, 1]
        np.testing.assert_allclose(eps, d[tag + "effective_permittivity"], rtol=1e-12)
        np.testing.assert_allclose(out.layers[i, :L, 2], d[tag + "ks"], rtol=1e-11)
        np.testing.assert_allclose(out.layers[i, :L, 3], d[tag + "ka"], rtol=1e-10)
        n_air = int(out.streams[i, 0])
        ang = np.rad2deg(np.arccos(out.streams[i, 1 : 1 + n_air]))
        np.testing.assert_allclose(ang, d[tag + "stream_angles"], rtol=1e-11)
        if tag + "streams_n" in d:
            assert list(out.layers[i, :L, 4].astype(int)) == list(d[tag + "streams_n"])


def test_albedo_above_one_sets_status(ctx):
    d = load_golden("dmrt_2layer_passive37")
    out = ctx.run(batch_from_fixture(d))
    assert out.status[0] == 3
    assert np.isnan(out.values).all()


def test_random_batch_against_oracle(ctx):
    """cfg2-like batch: every pair compared with the CPU oracle (sizes the oracle finishes in seconds)."""
    from oracle import dort_oracle as O
    from smrt_amd._native import PackedBatch

    rng = np.random.default_rng(7)
    S, L = 6, 20
    thick = np.concatenate([rng.uniform(0.05, 0.3, (S, L - 1)), np.full((S, 1), 100.0)], axis=1)
    dens = rng.uniform(150, 450, (S, L))
    temp = rng.uniform(230, 270, (S, L))
    lc = rng.uniform(5e-5, 3e-4, (S, L))
    freqs = np.array([10.65e9, 36.5e9, 89e9])
    theta = np.deg2rad([35.0, 55.0])
    batch = PackedBatch([L] * S, thick, dens / O.DENSITY_OF_ICE, temp, lc, None, freqs, theta)
    out = ctx.run(batch)
    assert (out.status == 0).all()
    for f, fr in enumerate(freqs):
        for s in range(S):
            sp = dict(thickness=thick[s], density=dens[s], temperature=temp[s], microstructure="exponential",
                      corr_length=lc[s])
            ref = O.solve(sp, fr, np.rad2deg(theta))
            assert np.abs(out.values[f * S + s] - ref).max() < TB_TOL


def test_cfg3_like_batch_64_streams(ctx):
    """BASELINE config 3 shape (DMRT-QCA short range, 50 layers, 64 streams, AMSR2 frequencies) on a small batch:
    every pair against the CPU oracle.  N = 128 > 64: work matrices live in the global workspace."""
    from oracle import dort_oracle as O
    from smrt_amd._native import PackedBatch

    rng = np.random.default_rng(3)
    S, L = 2, 50
    thick = np.concatenate([rng.uniform(0.05, 0.3, (S, L - 1)), np.full((S, 1), 100.0)], axis=1)
    dens = rng.uniform(150, 450, (S, L))
    temp = rng.uniform(230, 270, (S, L))
    rad = rng.uniform(5e-5, 1.5e-4, (S, L))
    stick = np.full((S, L), 0.2)
    freqs = np.array([6.925e9, 18.7e9, 89e9])
    batch = PackedBatch([L] * S, thick, dens / O.DENSITY_OF_ICE, temp, rad, stick, freqs, np.deg2rad([55.0]),
                        emmodel="dmrt_qca_shortrange", microstructure="sticky_hard_spheres", n_max_stream=64)
    out = ctx.run(batch)
    assert (out.status == 0).all()
    for f, fr in enumerate(freqs):
        for s in range(S):
            sp = dict(thickness=thick[s], density=dens[s], temperature=temp[s], microstructure="sticky_hard_spheres",
                      radius=rad[s], stickiness=stick[s])
            ref = O.solve(sp, fr, [55.0], emmodel="dmrt_qca_shortrange", n_max_stream=64)
            assert np.abs(out.values[f * S + s] - ref).max() < TB_TOL


def test_ragged_layers_and_pair_ranges(ctx):
    """Snowpacks with different layer counts in one batch; sub-ranges give the same numbers as the full range."""
    from oracle import dort_oracle as O
    from smrt_amd._native import PackedBatch

    rng = np.random.default_rng(11)
    S, Lmax = 5, 7
    nl = np.array([1, 7, 3, 2, 5], np.int32)
    thick = rng.uniform(0.05, 0.3, (S, Lmax))
    for s in range(S):
        thick[s, nl[s] - 1] = 50.0
    dens = rng.uniform(150, 450, (S, Lmax))
    temp = rng.uniform(230, 270, (S, Lmax))
    lc = rng.uniform(5e-5, 3e-4, (S, Lmax))
    freqs = np.array([18.7e9, 36.5e9])
    theta = np.deg2rad([55.0])
    batch = PackedBatch(nl, thick, dens / O.DENSITY_OF_ICE, temp, lc, None, freqs, theta, n_max_stream=16)
    full = ctx.run(batch)
    assert (full.status == 0).all()
    part = ctx.run(batch, pair_begin=3, pair_count=4)
    assert np.array_equal(part.values, full.values[3:7])
    for f, fr in enumerate(freqs):
        for s in range(S):
            n = nl[s]
            sp = dict(thickness=thick[s, :n], density=dens[s, :n], temperature=temp[s, :n],
                      microstructure="exponential", corr_length=lc[s, :n])
            ref = O.solve(sp, fr, [55.0], n_max_stream=16)
            assert np.abs(full.values[f * S + s] - ref).max() < TB_TOL


def test_full_size_batch_properties(ctx):
    """BASELINE config 2 size (1024 snowpacks x 5 frequencies): size-independent properties.
    (i) permutation equivariance: shuffling the snowpacks permutes the results bit-for-bit;
    (ii) determinism: two launches give identical bits; (iii) physical bounds: 0 < Tb < max layer temperature;
    (iv) isothermal non-scattering limit is covered in test_physics."""
    from smrt_amd._native import PackedBatch

    rng = np.random.default_rng(2)
    S, L = 1024, 20
    thick = np.concatenate([rng.uniform(0.05, 0.3, (S, L - 1)), np.full((S, 1), 100.0)], axis=1)
    dens = rng.uniform(150, 450, (S, L))
    temp = rng.uniform(230, 270, (S, L))
    lc = rng.uniform(5e-5, 3e-4, (S, L))
    freqs = np.array([10.65e9, 18.7e9, 23.8e9, 36.5e9, 89e9])
    theta = np.deg2rad([55.0])
    mk = lambda idx: PackedBatch([L] * S, thick[idx], dens[idx] / 916.7, temp[idx], lc[idx], None, freqs, theta)  # noqa
    ident = np.arange(S)
    a = ctx.run(mk(ident))
    b = ctx.run(mk(ident))
    assert (a.status == 0).all()
    assert np.array_equal(a.values, b.values)
    perm = rng.permutation(S)
    c = ctx.run(mk(perm))
    av = a.values.reshape(5, S, 2, 1)
    cv = c.values.reshape(5, S, 2, 1)
    assert np.array_equal(cv, av[:, perm])
    assert (a.values > 50).all() and (a.values < temp.max()).all()
