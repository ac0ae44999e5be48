"""The DEVICE code (smrt_amd/csrc/dort_*.hpp) executed on the CPU by the fiber emulator (tests/hostemu):
checks the kernel logic against the reference's golden vectors without a GPU, and that the result does not depend on
the order in which the emulated threads run between barriers (a missing barrier would)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from conftest import (COHERENT_FIXTURES, COHERENT_HOST_FIXTURES, HOST_EMMODEL_FIXTURES, ROUGH_SUBSTRATE_FIXTURES, ROUGH_SUBSTRATE_PASSIVE_FIXTURES, MIXED_FIXTURES, DENSE_AUTO_FIXTURES, WET_FIXTURES, MICRO_FIXTURES, IBA_FAMILY_FIXTURES, host_batch_from_fixture, PRUNE_ACTIVE_FIXTURES, PRUNE_FIXTURES, ROOT, SUBSTRATE_FIXTURES, assert_backscatter_close, load_golden, oracle_method_spread,
                      packed_batch_from_fixture, reference_method_spread)
from smrt_amd._native import PackedBatch, SmrtBatch

EMU_DIR = os.path.join(ROOT, "tests", "hostemu")
EMU_LIB = os.path.join(EMU_DIR, "libsmrt_emu.so")


@pytest.fixture(scope="module")
def emu():
    csrc = os.path.join(ROOT, "smrt_amd", "csrc")
    srcs = [os.path.join(EMU_DIR, "emu_lib.cpp"), os.path.join(EMU_DIR, "emu_runtime.hpp")] + sorted(
        os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith(".hpp"))
    if not os.path.exists(EMU_LIB) or any(os.path.getmtime(s) > os.path.getmtime(EMU_LIB) for s in srcs):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-I", EMU_DIR, "-o", EMU_LIB, srcs[0]])
    lib = C.CDLL(EMU_LIB)
    P = C.POINTER
    lib.smrt_emu_run.argtypes = [P(SmrtBatch), C.c_longlong, C.c_longlong, C.c_int, C.c_int, P(C.c_double),
                                 P(C.c_int32), P(C.c_double), P(C.c_double), P(C.c_double), P(C.c_long)]
    return lib


def run_fixture(lib, name, nt=64, order=0, freqs=None):
    d = load_golden(name)
    b = packed_batch_from_fixture(d, freqs)
    n = b.n_pairs
    out = np.empty((n,) + b.out_shape())
    st = np.empty(n, np.int32)
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))  # noqa: E731
    nb = C.c_long()
    rc = lib.smrt_emu_run(C.byref(b.struct), 0, n, nt, order, dp(out), st.ctypes.data_as(C.POINTER(C.c_int32)), None,
                          None, None, C.byref(nb))
    assert rc == 0
    ref = d["result"] if freqs is None else d["result"][freqs]
    return out, st, ref


@pytest.mark.parametrize("name", ["cfg1_iba_onelayer", "iba_2layer_passive37", "iba_L6_n8_angles",
                                  "iba_L3_n16_shallow", "dmrt_L8_n16", "dmrtcp_L5_n12", "dmrtcp_2layer_passive37"])
def test_emulated_kernel_matches_reference(emu, name):
    out, st, ref = run_fixture(emu, name)
    assert (st == 0).all()
    assert np.abs(out - ref).max() < 1e-6


@pytest.mark.parametrize("name,nt,order", [(n, (64, 128, 256, 64, 128)[i], i % 3) for i, n in enumerate(SUBSTRATE_FIXTURES)])
def test_emulated_kernel_substrate_atmosphere(emu, name, nt, order):
    """Flat / Reflector substrates (emitting or not) and the isotropic atmosphere on the device code."""
    out, st, ref = run_fixture(emu, name, nt=nt, order=order)
    assert (st == 0).all()
    assert np.abs(out - ref).max() < 1e-6


@pytest.mark.parametrize("name,nt,pipeline", [(PRUNE_FIXTURES[0], 64, 1), (PRUNE_FIXTURES[1], 128, 2),
                                              (PRUNE_FIXTURES[2], 256, 1), (PRUNE_FIXTURES[3], 128, 1),
                                              (PRUNE_ACTIVE_FIXTURES[0], 64, 1)])
def test_emulated_kernel_prune_deep_snowpack(emu, name, nt, pipeline):
    """DORT option prune_deep_snowpack on the device code (three-kernel pipelines: two-slot and four-slot finish)."""
    C.c_int.in_dll(emu, "smrt_emu_pipeline").value = pipeline
    try:
        out, st, ref = run_fixture(emu, name, nt=nt)
    finally:
        C.c_int.in_dll(emu, "smrt_emu_pipeline").value = 1
    assert (st == 0).all()
    if name in PRUNE_ACTIVE_FIXTURES:
        assert_backscatter_close(out, ref, spread=reference_method_spread(load_golden(name)))
    else:
        assert np.abs(out - ref).max() < 1e-6


REG_FIXTURES = ["cfg1_iba_onelayer", "iba_2layer_passive37", "iba_L6_n8_angles", "iba_L3_n16_shallow", "dmrt_L8_n16",
                "dmrtcp_L5_n12", "mixed_L4_n16_passive", "nonscattering_L3_n10_substrate"] + SUBSTRATE_FIXTURES + PRUNE_FIXTURES


@pytest.mark.parametrize("name,order", [(n, i % 3) for i, n in enumerate(dict.fromkeys(REG_FIXTURES))])
def test_emulated_register_resident_finish_kernel(emu, name, order):
    """The register-resident finish kernel (one wavefront per pair, pivot-free admittance recursion in MFMA register
    layout, dort_finish_reg.hpp) behind the same prep and Jacobi kernels: reference fixtures incl. substrates, atmosphere,
    pruning and heterogeneous snowpacks, in three fiber orders."""
    d = load_golden(name)
    if str(d["mode"]) != "P" or "coherent" in name or "rough" in name:
        pytest.skip("passive, Flat interfaces only")
    C.c_int.in_dll(emu, "smrt_emu_pipeline").value = 3
    try:
        out, st, ref = run_fixture(emu, name, nt=256, order=order)
    finally:
        C.c_int.in_dll(emu, "smrt_emu_pipeline").value = 1
    assert (st == 0).all()
    assert np.abs(out - ref).max() < 1e-6


def test_emulated_register_resident_finish_is_schedule_independent(emu):
    outs = []
    C.c_int.in_dll(emu, "smrt_emu_pipeline").value = 3
    try:
        for order in (0, 1, 2):
            outs.append(run_fixture(emu, "iba_L3_n16_substrate_atmosphere", nt=256, order=order)[0])
    finally:
        C.c_int.in_dll(emu, "smrt_emu_pipeline").value = 1
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])


STRIP_FIXTURES = ["cfg1_iba_onelayer", "iba_L6_n8_angles", "dmrt_L8_n16", "mixed_L4_n16_passive", "iba_L3_n16_substrate_atmosphere",
                  "dmrt_L4_n12_reflector", "iba_L2_n10_mirror_atmosphere_only", "iba_L8_n12_prune", "dmrt_L6_n10_prune_over_bad_layer"]


@pytest.mark.parametrize("name,order,pipeline", [(n, i % 3, 5 + (i + k) % 2) for i, n in enumerate(STRIP_FIXTURES) for k in (0, 1)
                                                 if k == 0 or i % 2 == 0])
def test_emulated_strip_finish_kernel(emu, name, order, pipeline):
    """The strip finish kernels (the pivot-free recursion on tile columns, one matrix at a time through LDS,
    dort_finish_strip.hpp) on the small reference fixtures: the eight-wavefront instance of the 64 < N <= 128 pipeline
    (the emulator's pipeline 5 forces that pipeline whatever the size) and the four-wavefront instance of the N <= 64
    pipeline (6) -- layers of 1 to 2 tiles, ragged stream counts, substrates, atmosphere, pruning, in three fiber orders.
    The eight-wavefront kernel's own size is covered by the configs[2] fixture below, the other's by the headline one."""
    C.c_int.in_dll(emu, "smrt_emu_pipeline").value = pipeline
    try:
        out, st, ref = run_fixture(emu, name, nt=256, order=order)
    finally:
        C.c_int.in_dll(emu, "smrt_emu_pipeline").value = 1
    assert (st == 0).all()
    assert np.abs(out - ref).max() < 1e-6


def test_emulated_strip_finish_kernel_at_its_own_size_and_any_schedule(emu):
    """50 layers x 64 streams (one frequency of the configs[2] fixture: up to 8 x 8 tiles) and schedule independence."""
    C.c_int.in_dll(emu, "smrt_emu_pipeline").value = 3
    try:
        out, st, ref = run_fixture(emu, "cfg3_dmrt_L50_n64_sp0", nt=256, order=1, freqs=[1])
        assert (st == 0).all() and np.abs(out - ref).max() < 1e-6
        C.c_int.in_dll(emu, "smrt_emu_pipeline").value = 6
        out, st, ref = run_fixture(emu, "cfg2_iba_L20_n32_sp1", nt=256, order=2, freqs=[0, 3])   # the headline shape: 3 - 4 tiles
        assert (st == 0).all() and np.abs(out - ref).max() < 1e-6
        outs = {}
        for pipeline in (5, 6):
            C.c_int.in_dll(emu, "smrt_emu_pipeline").value = pipeline
            outs[pipeline] = [run_fixture(emu, "iba_L3_n16_flat_substrate", nt=256, order=o)[0] for o in (0, 1, 2)]
    finally:
        C.c_int.in_dll(emu, "smrt_emu_pipeline").value = 1
    for o in outs.values():
        assert np.array_equal(o[0], o[1]) and np.array_equal(o[0], o[2])


@pytest.mark.parametrize("name,nt,pipeline,order", [("rough_iem_surface_L3_n10_passive", 256, 1, 0), ("rough_iem_inner_L3_n10_passive", 64, 0, 1),
                                                    ("rough_go_surface_L3_n10_active", 256, 1, 2), ("rough_iem_inner_L3_n10_active", 64, 0, 0)])
def test_emulated_kernel_rough_interfaces(emu, name, nt, pipeline, order):
    """SMRT_INTERFACE_HOST on the device code: dense interface matrices (inputs of the reference fixtures) composed with the
    layers below, passive and active, pipeline and fused kernels."""
    C.c_int.in_dll(emu, "smrt_emu_pipeline").value = pipeline
    try:
        out, st, ref = run_fixture(emu, name, nt=nt, order=order)
    finally:
        C.c_int.in_dll(emu, "smrt_emu_pipeline").value = 1
    assert (st == 0).all()
    if "active" in name:
        assert_backscatter_close(out, ref)
    else:
        assert np.abs(out - ref).max() < 1e-6


def test_emulated_kernel_is_schedule_independent(emu):
    base, _, _ = run_fixture(emu, "iba_L6_n8_angles", nt=128, order=0)
    for order in (1, 2):
        other, _, _ = run_fixture(emu, "iba_L6_n8_angles", nt=128, order=order)
        assert np.array_equal(base, other)


def test_emulated_kernel_full_size_pair(emu):
    """One 20-layer, 32-stream pair of the headline configuration at 89 GHz (strongest scattering)."""
    out, st, ref = run_fixture(emu, "cfg2_iba_L20_n32_sp1", nt=64, order=1, freqs=[5])
    assert (st == 0).all() and np.abs(out - ref).max() < 1e-6


@pytest.mark.parametrize("name,nt,order", [("cfg4_iba_active_L5_n16", 64, 0), ("dmrt_active_L3_n12", 128, 1),
                                           ("iba_shs_active_L3_n8", 64, 2), ("iba_active_L3_n10_m1_steep", 256, 1),
                                           ("iba_2layer_active19", 64, 0), ("iba_active_L3_n12_flat_substrate", 128, 2)])
def test_emulated_active_kernel_matches_reference(emu, name, nt, order):
    """Active mode (three polarisations, azimuth modes 0..m_max, coherent subtraction, backscatter read-out);
    the last case has N = 3 x 32 = 96 rows, i.e. the global-workspace variant of the kernel."""
    out, st, ref = run_fixture(emu, name, nt=nt, order=order)
    assert (st == 0).all()
    assert_backscatter_close(out, ref, spread=reference_method_spread(load_golden(name)))


def test_emulated_active_kernel_high_azimuth_order(emu):
    """m_max = 16 (512 azimuth samples) as in smrt/rtsolver/test_dort.py:13-38, at 8 streams, against the oracle."""
    from oracle import dort_oracle as O

    sp = dict(thickness=np.array([1000.0]), density=np.array([280.0]), temperature=np.array([265.0]),
              microstructure="exponential", corr_length=np.array([0.05e-3]))
    th = np.array([50.0])
    kw = dict(mode="A", theta_inc_deg=th, n_max_stream=8, m_max=16)
    ref = O.solve(sp, 10e9, th, method="schur_forcedtriu", **kw)
    b = PackedBatch([1], sp["thickness"], sp["density"] / 916.7, sp["temperature"], sp["corr_length"], None, [10e9],
                    np.deg2rad(th), emmodel="iba", microstructure="exponential", mode="A", n_max_stream=8, m_max=16)
    out = np.empty((1,) + b.out_shape())
    st = np.empty(1, np.int32)
    nb = C.c_long()
    rc = emu.smrt_emu_run(C.byref(b.struct), 0, 1, 64, 1, out.ctypes.data_as(C.POINTER(C.c_double)),
                          st.ctypes.data_as(C.POINTER(C.c_int32)), None, None, None, C.byref(nb))
    assert rc == 0 and st[0] == 0
    assert_backscatter_close(out[0], ref, spread=oracle_method_spread(sp, 10e9, th, ref, **kw))


@pytest.mark.parametrize("order", [0, 2])
def test_emulated_big_pipeline_against_the_oracle(emu, order):
    """The N > 128 pipeline (k_gmem_split_big.hip: CH = 4 row chunks, 512 threads) on a two-layer passive pair with 66
    streams -- N = 132, nine blocks of 16 with a ragged last one: the grouped Gauss-Jordan updates (two blocks per pass,
    last group of one), the LDS-staged operand tiles of the row-block products, the triangular product / solve with three
    column tiles per wavefront, the solver scratch of whole 16 x 16 blocks -- under two fiber schedules."""
    from oracle import dort_oracle as O

    sp = dict(thickness=np.array([0.15, 20.0]), density=np.array([230.0, 380.0]), temperature=np.array([255.0, 268.0]),
              microstructure="exponential", corr_length=np.array([2.2e-4, 0.9e-4]))
    th = np.array([30.0, 50.0])
    ref = O.solve(sp, 36.5e9, th, n_max_stream=66, method="schur_forcedtriu")
    b = PackedBatch([2], sp["thickness"], sp["density"] / 916.7, sp["temperature"], sp["corr_length"], None, [36.5e9],
                    np.deg2rad(th), emmodel="iba", microstructure="exponential", mode="P", n_max_stream=66)
    out = np.empty((1,) + b.out_shape())
    st = np.empty(1, np.int32)
    nb = C.c_long()
    rc = emu.smrt_emu_run(C.byref(b.struct), 0, 1, 256, order, out.ctypes.data_as(C.POINTER(C.c_double)),
                          st.ctypes.data_as(C.POINTER(C.c_int32)), None, None, None, C.byref(nb))
    assert rc == 0 and st[0] == 0
    assert np.abs(out[0] - ref).max() < 1e-7


def test_emulated_kernel_flags_albedo_above_one(emu):
    out, st, _ = run_fixture(emu, "dmrt_2layer_passive37")
    assert st[0] == 3 and np.isnan(out).all()


@pytest.mark.parametrize("name,freqs,min_fast", [("cfg2_iba_L20_n32_sp0", [4], 0.9), ("iba_L6_n8_angles", None, 0.9),
                                                 ("cfg4_iba_active_L5_n16", None, 0.6), ("dmrt_active_L3_n12", None, 0.0)])
def test_gauss_jordan_takes_its_pivots_from_the_diagonal_blocks(name, freqs, min_fast):
    """The optional fast panel of the Gauss-Jordan solves (-DSMRT_GJ_FAST_PANEL builds; pivots from the 16 x 16 diagonal
    block, dort_gauss_jordan.hpp:gj_panel16_fast, with the eigenpairs sorted by the Jacobi kernel): it is the one that
    runs; where its acceptance test refuses a block -- DMRT in active mode: clusters of equal eigenvalues -- the
    full-pivot panel takes over and the answer still matches the reference.  (The default build keeps the full-pivot
    panel: the fast one brought no speed-up inside the kernel, profiles/r2_gj_fast_panel.txt.)"""
    lib_path = os.path.join(EMU_DIR, "libsmrt_emu_fastpanel.so")
    csrc = os.path.join(ROOT, "smrt_amd", "csrc")
    srcs = [os.path.join(EMU_DIR, "emu_lib.cpp"), os.path.join(EMU_DIR, "emu_runtime.hpp")] + sorted(
        os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith(".hpp"))
    if not os.path.exists(lib_path) or any(os.path.getmtime(s) > os.path.getmtime(lib_path) for s in srcs):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-DSMRT_GJ_FAST_PANEL", "-I", EMU_DIR, "-o",
                               lib_path, srcs[0]])
    emu = C.CDLL(lib_path)
    P = C.POINTER
    emu.smrt_emu_run.argtypes = [P(SmrtBatch), C.c_longlong, C.c_longlong, C.c_int, C.c_int, P(C.c_double),
                                 P(C.c_int32), P(C.c_double), P(C.c_double), P(C.c_double), P(C.c_long)]
    counts = (C.c_long * 2)()
    emu.smrt_emu_panel_counts(counts)             # reset
    out, st, ref = run_fixture(emu, name, nt=64, freqs=freqs)
    emu.smrt_emu_panel_counts(counts)
    fast, slow = counts[0], counts[1]
    assert (st == 0).all() and fast + slow > 0
    assert fast / (fast + slow) >= min_fast, (fast, slow)
    if str(load_golden(name)["mode"]) == "A":
        assert_backscatter_close(out, ref, spread=reference_method_spread(load_golden(name)))
    else:
        assert np.abs(out - ref).max() < 1e-6


@pytest.mark.parametrize("n_max_stream,N,order", [(70, 97, 1), (70, 24, 1)])   # (the GPU suite runs the larger sizes end to end)
def test_blocked_jacobi_kernel_gives_the_singular_values(emu, n_max_stream, N, order):
    """The Jacobi kernel of the N > 128 pipeline (dort_jacobi_big.hpp: matrix in global memory, two column blocks at a
    time in LDS) on a random matrix: B' = B V has orthogonal columns, V is orthogonal, the column norms are the singular
    values -- whatever the order in which the emulated threads run between barriers."""
    emu.smrt_emu_jacobi_big.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int,
                                        C.c_double, C.c_double]
    NMAX = 2 * n_max_stream
    LD = (NMAX + 1) | 1
    rng = np.random.default_rng(N)
    A = rng.standard_normal((N, N)) @ np.diag(np.linspace(1, 30, N))
    buf = np.full((NMAX, LD), np.nan)
    buf[:N, :N] = A.T                       # column c of the matrix at buf[c, :N]
    sig = np.zeros(NMAX)
    rc = emu.smrt_emu_jacobi_big(n_max_stream, 2, N, buf.ctypes.data_as(C.POINTER(C.c_double)),
                                 sig.ctypes.data_as(C.POINTER(C.c_double)), order, 1e-26, 1e-15)
    assert rc == N
    Bp = buf[:N, :N].T
    V = np.linalg.solve(A, Bp)
    assert np.abs(V.T @ V - np.eye(N)).max() < 1e-11
    G = Bp.T @ Bp
    d = np.sqrt(np.diag(G))
    assert np.abs(G / np.outer(d, d) - np.eye(N)).max() < 1e-12
    sv = np.linalg.svd(A, compute_uv=False)
    np.testing.assert_allclose(np.sort(sig[:N])[::-1], sv, rtol=1e-12)


@pytest.mark.parametrize("name,nt", [(MIXED_FIXTURES[0], 64), (MIXED_FIXTURES[1], 256), (DENSE_AUTO_FIXTURES[0], 256),
                                     (DENSE_AUTO_FIXTURES[1], 128), (WET_FIXTURES[0], 256), (WET_FIXTURES[1], 64), (WET_FIXTURES[2], 128), (MICRO_FIXTURES[0], 256), (MICRO_FIXTURES[1], 64), (MICRO_FIXTURES[2], 256), (MICRO_FIXTURES[3], 64),
                                     (IBA_FAMILY_FIXTURES[0], 256), (IBA_FAMILY_FIXTURES[1], 64), (IBA_FAMILY_FIXTURES[2], 64),
                                     (IBA_FAMILY_FIXTURES[3], 256), (IBA_FAMILY_FIXTURES[4], 64), (IBA_FAMILY_FIXTURES[5], 64)])
def test_emulated_kernel_heterogeneous_snowpacks(emu, name, nt):
    """Per-layer emmodel and microstructure codes (smrt_batch.layer_kind) through the device code; IBA on the inverted
    medium for layers above half ice (SMRT_EM_IBA_INVERTED, the reference's dense_snow_correction="auto")."""
    out, st, ref = run_fixture(emu, name, nt=nt)
    assert (st == 0).all()
    if name.endswith("active"):
        assert_backscatter_close(out, ref, spread=reference_method_spread(load_golden(name)))
    else:
        assert np.abs(out - ref).max() < 1e-6


@pytest.mark.parametrize("name,nt,order", [(HOST_EMMODEL_FIXTURES[0], 64, 0), (HOST_EMMODEL_FIXTURES[1], 256, 1),
                                           (HOST_EMMODEL_FIXTURES[2], 128, 2), (COHERENT_HOST_FIXTURES[0], 256, 0),
                                           (COHERENT_HOST_FIXTURES[1], 64, 1)])
def test_emulated_kernel_with_host_evaluated_emmodels(emu, name, nt, order):
    """Emmodels without a device implementation (the reference's rayleigh and prescribed_kskaeps): the product
    evaluates the emmodel protocol on the host (DORT._evaluate_on_host), the device code takes ks / ka / permittivity /
    phase-matrix modes as numbers (SMRT_EM_HOST) and must reproduce the reference."""
    d = load_golden(name)
    b = host_batch_from_fixture(d)
    n = b.n_pairs
    out = np.empty((n,) + b.out_shape())
    st = np.empty(n, np.int32)
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))  # noqa: E731
    nb = C.c_long()
    rc = emu.smrt_emu_run(C.byref(b.struct), 0, n, nt, order, dp(out), st.ctypes.data_as(C.POINTER(C.c_int32)), None,
                          None, None, C.byref(nb))
    assert rc == 0 and (st == 0).all()
    if str(d["mode"]) == "A":
        assert_backscatter_close(out, d["result"], spread=reference_method_spread(d))
    else:
        assert np.abs(out - d["result"]).max() < 1e-6
    # a stream count that differs from the device's own is refused, not silently used
    b.host_streams[0, 0] += 1
    rc = emu.smrt_emu_run(C.byref(b.struct), 0, 1, nt, order, dp(out), st.ctypes.data_as(C.POINTER(C.c_int32)), None,
                          None, None, C.byref(nb))
    assert rc == 0 and st[0] == 5


@pytest.mark.parametrize("emmodel,micro,npol,m_max", [("iba", "exponential", 3, 2), ("iba", "sticky_hard_spheres", 3, 3),
                                                      ("iba", "exponential", 2, 0), ("dmrt_qca_shortrange", "sticky_hard_spheres", 3, 3),
                                                      ("nonscattering", "exponential", 2, 1)])
def test_device_ft_even_phase_against_the_oracle(emu, emmodel, micro, npol, m_max):
    """The emmodel protocol's ft_even_phase evaluated by the device function (dort_phase_kernel.hpp) on arbitrary
    cosine grids -- both hemispheres -- against the oracle's restatement of smrt/emmodel/common.py:56-131,349-399 and
    rayleigh.py:52-127 (itself pinned by the A-matrix stage fixtures of the reference)."""
    from oracle import dort_oracle as O
    from smrt_amd._native import EM_CODES, MS_CODES

    emu.smrt_emu_ft_even_phase.argtypes = [C.c_int, C.c_int] + [C.c_double] * 5 + [C.POINTER(C.c_double), C.c_int,
                                           C.POINTER(C.c_double), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
    mu_s = np.array([0.95, 0.6, 0.2, -0.3, -0.85])
    mu_i = np.array([0.9, 0.45, -0.1, -0.7])
    f, fv, T = 36.5e9, 0.3, 262.0
    mp = dict(corr_length=1.8e-4) if micro == "exponential" else dict(radius=1.4e-4, stickiness=0.2)
    layer = O.make_layers(emmodel, f, dict(thickness=[1.0], frac_volume=[fv], temperature=[T], microstructure=micro,
                                          **{k: [v] for k, v in mp.items()}))[0]
    ref = layer.ft_even_phase(mu_s, mu_i, m_max, npol)
    out = np.empty((npol, npol, m_max + 1, len(mu_s), len(mu_i)))
    p1, p2 = (mp["corr_length"], 0.0) if micro == "exponential" else (mp["radius"], mp["stickiness"])
    rc = emu.smrt_emu_ft_even_phase(EM_CODES[emmodel], MS_CODES[micro], f, fv, T, p1, p2,
                                    mu_s.ctypes.data_as(C.POINTER(C.c_double)), len(mu_s),
                                    mu_i.ctypes.data_as(C.POINTER(C.c_double)), len(mu_i), m_max, npol,
                                    out.ctypes.data_as(C.POINTER(C.c_double)))
    assert rc == 0
    np.testing.assert_allclose(out, np.asarray(ref), rtol=1e-11, atol=1e-14 * max(np.abs(ref).max(), 1e-300))


@pytest.mark.parametrize("name,nt,pipeline,order", [(COHERENT_FIXTURES[0], 64, 1, 0), (COHERENT_FIXTURES[0], 256, 0, 1),
                                                    (COHERENT_FIXTURES[1], 128, 1, 2)])
def test_emulated_kernel_process_coherent_layers(emu, name, nt, pipeline, order):
    """DORT option process_coherent_layers on the device code: the 2 mm crust and the 3 mm ice lens of the fixture are
    taken out -- both at the lower frequencies, one at 36.5 GHz -- and replaced by coherent interfaces; pipeline and
    fused kernels, passive and active, against the reference."""
    C.c_int.in_dll(emu, "smrt_emu_pipeline").value = pipeline
    try:
        out, st, ref = run_fixture(emu, name, nt=nt, order=order)
    finally:
        C.c_int.in_dll(emu, "smrt_emu_pipeline").value = 1
    assert (st == 0).all()
    d = load_golden(name)
    if str(d["mode"]) == "A":
        assert_backscatter_close(out, ref, spread=reference_method_spread(d))
    else:
        assert np.abs(out - ref).max() < 1e-6


@pytest.mark.parametrize("name,nt,pipeline,order", [(ROUGH_SUBSTRATE_FIXTURES[0], 256, 1, 0), (ROUGH_SUBSTRATE_FIXTURES[1], 64, 0, 1)])
def test_emulated_kernel_rough_substrate(emu, name, nt, pipeline, order):
    """SMRT_SUBSTRATE_HOST: the bottom-up recursion of every azimuth mode starts from the dense reflection matrix of a rough
    substrate handed over by the caller (here: what the reference's geometrical_optics / iem_fung92 substrates gave),
    pipeline and fused active kernels, against the reference's backscatter."""
    C.c_int.in_dll(emu, "smrt_emu_pipeline").value = pipeline
    try:
        out, st, ref = run_fixture(emu, name, nt=nt, order=order)
    finally:
        C.c_int.in_dll(emu, "smrt_emu_pipeline").value = 1
    assert (st == 0).all()
    assert_backscatter_close(out, ref, spread=reference_method_spread(load_golden(name)))


@pytest.mark.parametrize("name,nt,pipeline", [(ROUGH_SUBSTRATE_PASSIVE_FIXTURES[0], 256, 1), (ROUGH_SUBSTRATE_PASSIVE_FIXTURES[1], 64, 2),
                                              (ROUGH_SUBSTRATE_PASSIVE_FIXTURES[1], 128, 0)])
def test_emulated_kernel_rough_substrate_passive(emu, name, nt, pipeline):
    """SMRT_SUBSTRATE_HOST in passive mode: reflection matrix of mode 0 and emissivity diagonal from the caller."""
    C.c_int.in_dll(emu, "smrt_emu_pipeline").value = pipeline
    try:
        out, st, ref = run_fixture(emu, name, nt=nt)
    finally:
        C.c_int.in_dll(emu, "smrt_emu_pipeline").value = 1
    assert (st == 0).all()
    assert np.abs(out - ref).max() < 1e-6


def _own_rough_option_pack(case):
    """smrt_amd's own objects for one of conftest.ROUGH_OPTION_CASES: the rough interface / substrate by name."""
    from smrt_amd import make_interface, make_snowpack, make_soil, sensor_list

    itf = ["flat"] * len(case["thickness"])
    if case.get("interface"):
        model, kw, where = case["interface"]
        itf[where] = make_interface(model, **kw)
    substrate = None
    if case.get("substrate"):
        model, kw = case["substrate"]
        substrate = make_soil(model, complex(*case["substrate_eps"]), case["substrate_temperature"], **kw)
    pack = make_snowpack(case["thickness"], "exponential", density=case["density"], temperature=case["temperature"],
                         corr_length=case["corr_length"], interface=itf, substrate=substrate)
    sensor = (sensor_list.active if case["mode"] == "A" else sensor_list.passive)(case["frequency"], case["theta"])
    return sensor, pack


def check_rough_option_case(name, case):
    """Model.run on smrt_amd's own objects against the reference's result for the same models and options; where the
    reference's answer depends on the rough model at all, the answer with Flat interfaces must NOT pass."""
    import warnings

    from smrt_amd import make_model

    d = load_golden(name)
    sensor, pack = _own_rough_option_pack(case)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        res = make_model("iba", "dort", rtsolver_options=case["options"]).run(sensor, pack)
    got, ref, flat = np.asarray(res.data.values), d["result"][0], d["result_flat"][0]
    assert got.shape == ref.shape
    if case["mode"] == "P":
        assert np.abs(got - ref).max() < 1e-6
        if np.abs(ref - flat).max() > 1e-5:
            assert np.abs(got - flat).max() > 1e-5
    else:
        assert_backscatter_close(got, ref)
        rel = (np.abs(ref - flat) / np.abs(ref))[:2, :2].max()
        if rel > 1e-7:
            assert (np.abs(got - flat) / np.abs(ref))[:2, :2].max() > 1e-7
    assert len(np.ravel(res.other_data["ks"].values)) == int(d["kept_layers"])


@pytest.mark.parametrize("name", ["rough_prune_iem_L4_n10_passive", "rough_prune_go_L4_n10_active",
                                  "rough_coherent_iem_L5_n10_passive", "rough_coherent_adjacent_L5_n10_passive",
                                  "rough_coherent_gosub_L4_n10_active"])
def test_emulated_rough_interfaces_under_prune_and_coherent_options(name, emulated):
    """prune_deep_snowpack cutting above a rough interface (its dense reflection closes the recursion, dort.py:443-452) and
    process_coherent_layers with a rough interface elsewhere / on the collapsed layer / with a rough substrate (matrices
    sampled on the streams of the reduced snowpack; coherent_flat.py:16-57): the whole product path -- own interface
    evaluators, host packing, device source under the emulator -- against the reference (fixtures of
    tests/golden/make_rough_option_fixtures.py)."""
    from conftest import ROUGH_OPTION_CASES

    check_rough_option_case(name, ROUGH_OPTION_CASES[name])


def _eig_item(emu, NMAX, N, A, order):
    emu.smrt_emu_eig_item.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_longlong)]
    LD = (NMAX + 1) | 1
    buf = np.full((NMAX, LD), np.nan)
    buf[:N, :N] = A.T                       # column c of the matrix at buf[c, :N]
    sig = np.zeros(NMAX)
    nrec = C.c_longlong()
    rc = emu.smrt_emu_eig_item(NMAX, N, buf.ctypes.data_as(C.POINTER(C.c_double)), sig.ctypes.data_as(C.POINTER(C.c_double)), order,
                               C.byref(nrec))
    return rc, buf[:N, :N].T.copy(), sig[:N].copy(), nrec.value


@pytest.mark.parametrize("NMAX,N,order", [(64, 48, 0), (64, 64, 1), (64, 57, 2), (64, 33, 1), (48, 40, 0), (21, 21, 2), (64, 7, 0),
                                          (64, 3, 1), (64, 2, 0), (64, 1, 0)])
def test_symmetric_eigensolver_kernels(emu, NMAX, N, order):
    """The symmetric eigensolver of the N <= 64 pipelines (dort_eig_kernel.hpp: gram -> tridiag -> chase -> vectors, as
    k_eig.hip launches them) on one matrix B: the columns it returns are B' = U Sigma with U the eigenvectors of B B^T --
    orthonormal to rounding, the residual of the eigen-equation at rounding level, sigma the singular values of B --
    whatever the order in which the emulated lanes run between their rendezvous, for every padded size and for staging
    layouts whose leading dimension is not a multiple of eight."""
    rng = np.random.default_rng(100 * NMAX + N)
    A = rng.standard_normal((N, N)) @ np.diag(np.linspace(1, 30, N)) * 1e-3    # graded columns, small units
    rc, Bp, sig, nrec = _eig_item(emu, NMAX, N, A, order)
    assert rc == N
    sv = np.linalg.svd(A, compute_uv=False)
    U = Bp / sig[None, :]
    assert np.abs(U.T @ U - np.eye(N)).max() < 2e-14
    assert np.abs(A @ A.T @ U - U * sig[None, :] ** 2).max() < 5e-15 * sv.max() ** 2
    # (the price of the squared problem: eps * condition^2 on the small singular values)
    np.testing.assert_allclose(np.sort(sig)[::-1], sv, rtol=1e-16 * (sv.max() / sv.min()) ** 2 + 1e-13)
    if N > 8:
        assert 0.5 * N * N < nrec < 1.5 * N * N      # ~0.85 N^2 plane rotations + the identity padding of the groups


def test_symmetric_eigensolver_on_degenerate_and_diagonal_matrices(emu):
    """Inputs a QL iteration can stumble over: an exactly diagonal B (no scattering: every reflector is skipped, every
    eigenvalue stands alone), exactly repeated singular values (V / H pairs of a non-scattering layer), a rank-one
    perturbation of the identity, and singular values spread over eight decades."""
    N, NMAX = 24, 32
    rng = np.random.default_rng(3)
    Q1, _ = np.linalg.qr(rng.standard_normal((N, N)))
    Q2, _ = np.linalg.qr(rng.standard_normal((N, N)))
    cases = {
        "diagonal": np.diag(np.linspace(0.5, 40.0, N)),
        "pairs": Q1 @ np.diag(np.repeat(np.linspace(1.0, 12.0, N // 2), 2)) @ Q2,
        "rank one": np.eye(N) + 0.3 * np.outer(np.ones(N), np.ones(N)) / N,
        "graded": Q1 @ np.diag(np.logspace(0, -4, N)) @ Q2,
    }
    for name, A in cases.items():
        rc, Bp, sig, _ = _eig_item(emu, NMAX, N, A, 0)
        assert rc == N, name
        sv = np.linalg.svd(A, compute_uv=False)
        U = Bp / sig[None, :]
        assert np.abs(U.T @ U - np.eye(N)).max() < 2e-14, name
        assert np.abs(A @ A.T @ U - U * sig[None, :] ** 2).max() < 1e-14 * sv.max() ** 2, name
        np.testing.assert_allclose(np.sort(sig)[::-1], sv, rtol=0, atol=2e-15 * sv.max() ** 2 / sv.min(), err_msg=name)


@pytest.mark.parametrize("name", ["iba_2layer_passive37", "iba_L6_n8_angles", "dmrt_L8_n16"])
def test_jacobi_kernel_still_runs_the_passive_pipeline(emu, name):
    """The one-sided Jacobi kernel stays the diagonalisation of active mode and of N > 64; with smrt_emu_eig = 0 (the
    library's smrt_dort_set_diagonalisation(SMRT_DIAG_JACOBI)) it runs the passive N <= 64 pipeline too, and the two
    diagonalisations agree far inside the parity bar."""
    C.c_int.in_dll(emu, "smrt_emu_pipeline").value = 6
    try:
        sym, st_s, ref = run_fixture(emu, name)
        C.c_int.in_dll(emu, "smrt_emu_eig").value = 0
        jac, st_j, _ = run_fixture(emu, name)
    finally:
        C.c_int.in_dll(emu, "smrt_emu_eig").value = 1
        C.c_int.in_dll(emu, "smrt_emu_pipeline").value = 1
    assert (st_s == 0).all() and (st_j == 0).all()
    assert np.abs(sym - ref).max() < 1e-6 and np.abs(jac - ref).max() < 1e-6
    assert np.abs(sym - jac).max() < 1e-8


def test_emulator_under_address_sanitizer():
    """The device source under an AddressSanitizer build of the emulator (tests/hostemu/asan_cases.py): an active pair at
    N = 132 on the N > 128 pipeline and the symmetric eigensolver on odd staging layouts.  No GPU test can see a write
    into a NEIGHBOUR's workspace when there is no neighbour, or one that lands in padding; this build can (it found the
    solver-scratch overflow of round 5 and, in round 6, a reflector column written past the leading dimension)."""
    import sys

    asan_rt = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not asan_rt or not os.path.exists(asan_rt):
        pytest.skip("no AddressSanitizer runtime in this toolchain")
    lib_path = os.path.join(EMU_DIR, "libsmrt_emu_asan.so")
    csrc = os.path.join(ROOT, "smrt_amd", "csrc")
    srcs = [os.path.join(EMU_DIR, "emu_lib.cpp"), os.path.join(EMU_DIR, "emu_runtime.hpp")] + sorted(
        os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith(".hpp"))
    if not os.path.exists(lib_path) or any(os.path.getmtime(s) > os.path.getmtime(lib_path) for s in srcs):
        subprocess.check_call(["g++", "-O1", "-g", "-fsanitize=address", "-std=c++17", "-shared", "-fPIC", "-I", EMU_DIR, "-o", lib_path, srcs[0]])
    env = dict(os.environ, LD_PRELOAD=asan_rt, ASAN_OPTIONS="detect_leaks=0")
    out = subprocess.run([sys.executable, os.path.join(EMU_DIR, "asan_cases.py"), lib_path], capture_output=True, text=True, env=env,
                         timeout=1500)
    assert out.returncode == 0 and out.stdout.strip().endswith("asan cases ok"), (out.stdout[-500:], out.stderr[-3000:])
    assert "ERROR: AddressSanitizer" not in out.stderr


@pytest.mark.parametrize("n,ke,pa,order", [(8, 1.3, 0.9, 0), (6, 2.0, 0.05, 1), (12, 0.02, 0.01, 2), (16, 0.02, 1e-6, 0), (64, 3.0, 0.2, 1),
                                           (32, 40.0, 1e-9, 2), (5, 1.0, 0.0, 0)])
def test_rayleigh_closed_form_kernel(emu, n, ke, pa, order):
    """dort_rayleigh_kernel.hpp on one synthetic layer: D X+ D = diag(a) - Y Y^T (every pole twice, rank-two update) from the
    2 x 2 secular problem.  V = D^-1 A+ is orthogonal to rounding, the eigen-equation holds to rounding, sigma^2 are the
    eigenvalues -- for strong, weak (roots 1e-12 of an interval away from their pole) and no scattering, up to N = 128."""
    N, NMAX = 2 * n, 128
    LD = (NMAX + 1) | 1
    mu = 0.98 - 0.93 * np.arange(n) / n
    u = 0.3 + 0.01 * np.arange(N)
    Ap = np.full((NMAX, LD), np.nan); sig = np.zeros(NMAX); inv_d2 = np.zeros(NMAX)
    dp = lambda x: np.ascontiguousarray(x).ctypes.data_as(C.POINTER(C.c_double))  # noqa: E731
    emu.smrt_emu_rayleigh_item.argtypes = [C.c_int, C.c_int, C.c_double, C.c_double] + [C.POINTER(C.c_double)] * 5 + [C.c_int]
    rc = emu.smrt_emu_rayleigh_item(NMAX, n, ke, pa, dp(mu), dp(u), dp(Ap), dp(sig), dp(inv_d2), order)
    assert rc == N
    D = np.sqrt(ke / np.repeat(mu, 2))
    u1 = np.empty(N); u1[0::2] = mu ** 2; u1[1::2] = 1.0
    u2 = np.zeros(N); u2[0::2] = 1.0 - mu ** 2
    Y = np.column_stack([D * np.sqrt(0.5 * pa) * u * u1, D * np.sqrt(pa) * u * u2])
    M = np.diag(D ** 4) - Y @ Y.T
    V = Ap[:N, :N].T / D[:, None]
    lam = sig[:N] ** 2
    assert np.abs(V.T @ V - np.eye(N)).max() < 5e-13
    assert np.abs(M @ V - V * lam[None, :]).max() < 1e-13 * np.abs(lam).max()
    np.testing.assert_allclose(np.sort(lam), np.linalg.eigvalsh(M), rtol=1e-12)
    np.testing.assert_allclose(inv_d2[:N], 1.0 / D ** 2, rtol=1e-15)


def test_rayleigh_closed_form_kernel_flags_an_albedo_above_one(emu):
    """diag(a) - Y Y^T stops being positive definite exactly when the Cholesky factorisation of X+ would fail: status 3."""
    n, NMAX = 6, 32
    mu = np.linspace(0.95, 0.2, n)
    dp = lambda x: np.ascontiguousarray(x).ctypes.data_as(C.POINTER(C.c_double))  # noqa: E731
    emu.smrt_emu_rayleigh_item.argtypes = [C.c_int, C.c_int, C.c_double, C.c_double] + [C.POINTER(C.c_double)] * 5 + [C.c_int]
    Ap = np.zeros((NMAX, (NMAX + 1) | 1)); sig = np.zeros(NMAX); inv_d2 = np.zeros(NMAX)
    assert emu.smrt_emu_rayleigh_item(NMAX, n, 1.0, 30.0, dp(mu), dp(np.full(2 * n, 0.8)), dp(Ap), dp(sig), dp(inv_d2), 0) == -3


@pytest.mark.parametrize("name,pipeline", [("dmrt_L8_n16", 6), ("dmrtcp_L5_n12", 6), ("dmrt_L4_n12_reflector", 6), ("nonscattering_L3_n10_substrate", 6),
                                           ("dmrt_wet_L3_n12_passive", 6), ("dmrt_L8_n16", 5)])
def test_rayleigh_layers_with_and_without_the_closed_form(emu, name, pipeline):
    """Layers with a Rayleigh phase matrix through the strip pipelines (6: four wavefronts, N <= 64; 5: the global-workspace
    pipeline with the eight-wavefront strip kernel) with the closed-form kernel (the default) and with the Cholesky +
    diagonalisation route (smrt_emu_rayleigh = 0, the library's SMRT_DORT_RAYLEIGH=0): both at the reference's numbers,
    and with each other to 1e-9 K."""
    C.c_int.in_dll(emu, "smrt_emu_pipeline").value = pipeline
    try:
        closed, st_c, ref = run_fixture(emu, name, nt=256)
        C.c_int.in_dll(emu, "smrt_emu_rayleigh").value = 0
        chol, st_j, _ = run_fixture(emu, name, nt=256)
    finally:
        C.c_int.in_dll(emu, "smrt_emu_rayleigh").value = 1
        C.c_int.in_dll(emu, "smrt_emu_pipeline").value = 1
    assert (st_c == 0).all() and (st_j == 0).all()
    assert np.abs(closed - ref).max() < 1e-6 and np.abs(chol - ref).max() < 1e-6
    assert np.abs(closed - chol).max() < 1e-9


def test_strip_layer_step_on_thin_layers_next_to_thick_ones(emu):
    """The layer step of the strip finish kernels takes ONE inversion where every stream of the layer has sigma d >= 1e-3
    (Woodbury on M3: a difference of O(1 / (sigma d)) terms) and two elsewhere; every matrix is inverted scaled to unit
    pivots (dort_finish_strip.hpp; DESIGN 3d -- unscaled, the first build was wrong by 1e-5 K on exactly these media).
    0.1 mm ... 3 m layers at 89 - 183 GHz, thin ones on top of, between and under thick ones, against the oracle."""
    from oracle import dort_oracle as O
    rng = np.random.default_rng(3)
    theta = np.array([10.0, 55.0])
    C.c_int.in_dll(emu, "smrt_emu_pipeline").value = 6
    worst, checked = 0.0, 0
    try:
        for trial in range(14):
            L = int(rng.integers(2, 5))
            n_str = int(rng.choice([4, 7, 12]))
            thick = 10.0 ** rng.uniform(-4, 0.5, L)
            thick[int(rng.integers(0, L))] = 10.0 ** rng.uniform(-4, -3)   # at least one thin layer
            thick[-1] = rng.choice([0.3, 100.0])
            fv = rng.uniform(0.05, 0.49, L)
            temp = rng.uniform(200, 272.9, L)
            lc = 10.0 ** rng.uniform(-5, -3.2, L)
            f = float(rng.choice([89e9, 150e9, 183e9]))
            sp = dict(thickness=thick, frac_volume=fv, temperature=temp, microstructure="exponential", corr_length=lc)
            try:
                ref = O.solve(sp, f, theta, n_max_stream=n_str)
            except O.OracleError:
                continue
            b = PackedBatch([L], thick[None], fv[None], temp[None], lc[None], None, np.array([f]), np.deg2rad(theta), n_max_stream=n_str)
            out = np.empty((1,) + b.out_shape())
            st = np.empty(1, np.int32)
            nb = C.c_long()
            rc = emu.smrt_emu_run(C.byref(b.struct), 0, 1, 256, trial % 3, out.ctypes.data_as(C.POINTER(C.c_double)),
                                  st.ctypes.data_as(C.POINTER(C.c_int32)), None, None, None, C.byref(nb))
            assert rc == 0 and st[0] == 0
            worst = max(worst, float(np.abs(out[0] - ref).max()))
            checked += 1
    finally:
        C.c_int.in_dll(emu, "smrt_emu_pipeline").value = 1
    assert checked >= 10 and worst < 2e-9, (checked, worst)
