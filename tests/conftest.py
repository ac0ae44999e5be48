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
    sp["microstructure"] = str(d["microstructure"]) if np.ndim(d["microstructure"]) == 0 else [str(m) for m in d["microstructure"]]
    for k in ("corr_length", "radius", "stickiness", "repeat_distance", "porod_length", "polydispersity", "ks", "ka", "eps_re",
              "eps_im", "liquid_water"):
        if k in d:
            sp[k] = d[k]
    return sp


def fixture_options(d):
    o = dict(n_max_stream=int(d.get("opt_n_max_stream", 32)), m_max=int(d.get("opt_m_max", 2)))
    if "opt_prune_deep_snowpack" in d:
        v = np.asarray(d["opt_prune_deep_snowpack"])
        o["prune_deep_snowpack"] = 6.0 if v.dtype == bool else float(v)  # True means 6 (dort.py:176-177)
    return o


def fixture_coherent(d):
    return bool(d["opt_process_coherent_layers"]) if "opt_process_coherent_layers" in d else False


def fixture_substrate(d, i):
    """Substrate description of frequency index i of a fixture, in the oracle's form (None if there is none)."""
    if "substrate_kind" not in d:
        return None
    T = float(d["substrate_temperature"])
    sub = dict(kind=str(d["substrate_kind"]), temperature=None if np.isnan(T) else T)
    if sub["kind"] == "host":   # dense reflection matrices per azimuth mode, stored in the fixture (active mode)
        if str(d["mode"]) == "P":   # mode 0 only, plus the emissivity diagonal
            n = len(d["sub_emis"]) // 2
            sub["R"], sub["Rcoh"] = [d["sub_R_m0"]], [np.diag(d["sub_R_m0"])]
            sub["emissivity"] = d["sub_emis"].reshape(n, 2).T
            return sub
        modes = int(d["opt_m_max"]) + 1
        sub["R"] = [d["sub_R_m%d" % m] for m in range(modes)]
        sub["Rcoh"] = [d["sub_Rcoh_m%d" % m] for m in range(modes)]
        return sub
    if sub["kind"] == "flat":
        sub["eps"] = complex(d["substrate_eps"][i])
    else:
        sub["R"] = tuple(d["substrate_R"][i])
    return sub


def fixture_atmosphere(d, i):
    if "atm_tb_down" not in d:
        return None
    return dict(tb_down=float(d["atm_tb_down"][i]), tb_up=float(d["atm_tb_up"][i]),
                transmittance=float(d["atm_trans"][i]))


def packed_batch_from_fixture(d, freqs=None):
    """The one-snowpack PackedBatch (C-ABI input) of a fixture: all its frequencies, or the listed ones."""
    from smrt_amd._native import PackedBatch

    from smrt_amd._native import EM_CODES, MS_CODES

    sp = snowpack_dict(d)
    ms = sp["microstructure"]
    layer_kind = None
    emmodel = str(d["emmodel"]) if np.ndim(d["emmodel"]) == 0 else [str(e) for e in d["emmodel"]]
    if emmodel == "iba_dense_auto":   # what the product's host side does with that option (rtsolver/dort.py:_pack)
        dense = np.asarray(sp["frac_volume"]) > 0.5
        emmodel = ["iba_inverted" if x else "iba" for x in dense]
        sp["frac_volume"] = np.where(dense, 1.0 - np.asarray(sp["frac_volume"]), sp["frac_volume"])
    host_scalars = None
    family = ("iba_original", "iba_maxwell_garnett", "symsce_torquato21")     # IBA's phase function, other scalars: SMRT_EM_IBA_HOST
    if emmodel in family or (isinstance(emmodel, list) and any(e in family for e in emmodel)):
        # the scalars the emmodel OBJECT would hand over (rtsolver/dort.py:_iba_scalars_on_host), here from the oracle's
        # layer objects: ks, ka, effective permittivity, the coefficient of the phase matrix -- per frequency and layer
        from oracle import dort_oracle as O

        L = len(sp["thickness"])
        names = emmodel if isinstance(emmodel, list) else [emmodel] * L
        fsel = np.atleast_1d(d["frequency"])[slice(None) if freqs is None else freqs]
        hl, hc = np.zeros((len(fsel), 1, L, 4)), np.zeros((len(fsel), 1, L))
        for fi, f in enumerate(fsel):
            for l, em in enumerate(O.make_layers(names, float(f), sp)):
                hl[fi, 0, l] = em.ks, em.ka, complex(em.eps_eff).real, complex(em.eps_eff).imag
                hc[fi, 0, l] = em.iba_coeff
        host_scalars = (hl, hc)
        emmodel = ["iba_host" if e in family else e for e in names]
        if "symsce_torquato21" in names:   # ... at the complex wavenumber of the strong-contrast expansion: 4 + the model's code
            assert all(e == "symsce_torquato21" for e in names)
            from smrt_amd._native import MS_CODES as _MS
            from smrt_amd.core.layer import MICROSTRUCTURE_ARGS, device_microstructure_params
            per_layer = ms if isinstance(ms, list) else [ms] * L
            at = lambda k, l: float(np.nan_to_num(np.broadcast_to(sp[k], (L,))[l])) if k in sp else 1000.0   # noqa: E731  (stickiness default)
            pq = [device_microstructure_params(m, float(np.broadcast_to(sp["frac_volume"], (L,))[l]),
                                               **{k: at(k, l) for k in MICROSTRUCTURE_ARGS[m]}) for l, m in enumerate(per_layer)]
            sp = dict(sp, complex_k_p1=np.array([q[0] for q in pq]), complex_k_p2=np.array([q[1] for q in pq]))
            ms = [{0: "exponential_complex_k", 1: "sticky_hard_spheres_complex_k", 2: "independent_sphere_complex_k",
                   3: "teubner_strey_complex_k"}[_MS[m]] for m in per_layer]
    if isinstance(ms, list) or isinstance(emmodel, list):   # heterogeneous snowpack: per-layer codes and parameters
        L = len(sp["thickness"])
        msl = ms if isinstance(ms, list) else [ms] * L
        eml = emmodel if isinstance(emmodel, list) else [emmodel] * L
        layer_kind = [EM_CODES[e] + 16 * MS_CODES[m] for e, m in zip(eml, msl)]
        col = lambda k: np.nan_to_num(np.broadcast_to(sp.get(k, np.zeros(L)), (L,)).astype(float))  # noqa: E731
        # micro_p1 / micro_p2 of every layer through the product's own mapping (core/layer.py): corr_length | radius |
        # stickiness as they are, Teubner-Strey's repeat distance as Y, the unified parameters reparametrised
        from smrt_amd.core.layer import MICROSTRUCTURE_ARGS, device_microstructure_params
        fvl = np.broadcast_to(sp["frac_volume"], (L,))
        pp = [(float(col("complex_k_p1")[l]), float(col("complex_k_p2")[l])) if m.endswith("_complex_k") else
              device_microstructure_params(m, float(fvl[l]), **{a: float(col(a)[l]) for a in MICROSTRUCTURE_ARGS[m]})
              for l, m in enumerate(msl)]
        p1, p2 = np.array([q[0] for q in pp]), np.array([q[1] for q in pp])
        ms, emmodel = msl[0], eml[0]
        ms = "exponential" if ms.endswith("_complex_k") else ms   # (per-layer codes only: the batch-level one is not read)
    else:
        p1 = sp["corr_length"] if ms == "exponential" else sp["radius"]
        p2 = None if ms == "exponential" else np.broadcast_to(sp["stickiness"], p1.shape)
    sel = slice(None) if freqs is None else freqs
    o = fixture_options(d)
    active = str(d["mode"]) == "A"
    substrate = atmosphere = None
    if "substrate_kind" in d and str(d["substrate_kind"]) == "host":
        # rough substrate (active): the dense reflection matrices of the fixture, zero-padded to NE = 3 n_max_stream
        nm, ne = (o["m_max"] + 1 if active else 1), 3 * o["n_max_stream"]
        R, Rc = np.zeros((1, nm, ne, ne)), np.zeros((1, nm, ne))
        for m in range(nm):
            k = d["sub_R_m%d" % m].shape[0]
            R[0, m, :k, :k] = d["sub_R_m%d" % m]
            Rc[0, m, :k] = d["sub_Rcoh_m%d" % m] if active else d["sub_emis"]   # passive: the emissivity diagonal
        substrate = ("host", R, Rc) if active else ("host", R, Rc, [float(d["substrate_temperature"])])
    elif "substrate_kind" in d:
        kind = str(d["substrate_kind"])
        q = d["substrate_eps"][sel] if kind == "flat" else None
        q1 = q.real if kind == "flat" else d["substrate_R"][sel][:, 0]
        q2 = q.imag if kind == "flat" else d["substrate_R"][sel][:, 1]
        substrate = (kind, np.asarray(q1)[:, None], np.asarray(q2)[:, None], [float(d["substrate_temperature"])])
    if "atm_tb_down" in d:
        atmosphere = (d["atm_tb_down"][sel], d["atm_tb_up"][sel], d["atm_trans"][sel])
    return PackedBatch([len(sp["thickness"])], sp["thickness"], sp["frac_volume"], sp["temperature"], p1, p2,
                       d["frequency"][sel], np.deg2rad(d["theta_inc_deg"] if active else d["theta_deg"]),
                       emmodel=emmodel, microstructure=ms, mode="A" if active else "P",
                       n_max_stream=o["n_max_stream"], m_max=o["m_max"], substrate=substrate, atmosphere=atmosphere,
                       prune_deep_snowpack=o.get("prune_deep_snowpack"), layer_kind=layer_kind,
                       process_coherent_layers=fixture_coherent(d), liquid_water=sp.get("liquid_water"),
                       host_scalars=host_scalars,
                       host_interfaces=(pack_host_interfaces([fixture_interfaces(d)] * len(np.atleast_1d(d["frequency"][sel])),
                                                             len(sp["thickness"]), o["n_max_stream"],
                                                             (o["m_max"] + 1) if active else 1)
                                        if "rough_interface" in d else None))


SUBSTRATE_FIXTURES = ["iba_L3_n16_flat_substrate", "iba_L3_n16_substrate_atmosphere", "dmrt_L4_n12_reflector",
                      "iba_L2_n10_mirror_atmosphere_only", "nonscattering_L3_n10_substrate"]
PASSIVE_FIXTURES = [
    "cfg1_iba_onelayer",
    "iba_2layer_passive37",
    "cfg2_iba_L20_n32_sp0",
    "cfg2_iba_L20_n32_sp1",
    "iba_L6_n8_angles",
    "iba_L3_n16_shallow",
    "dmrt_L8_n16",
    "cfg3_dmrt_L50_n64_sp0",
    "cfg3_dmrt_L50_n64_amsr2_sp1",   # configs[2] at full size: 50 layers, 64 streams, all seven AMSR2 frequencies
    "dmrtcp_2layer_passive37",
    "dmrtcp_L5_n12",
]
ACTIVE_FIXTURES = ["iba_2layer_active19", "cfg4_iba_active_L5_n16", "iba_active_L4_n32_ku", "dmrt_active_L3_n12",
                   "iba_shs_active_L3_n8", "iba_active_L3_n10_m1_steep", "iba_active_L3_n12_flat_substrate"]
# DORT option prune_deep_snowpack: cut at layer 5 / 1 / not at all depending on the frequency; with a substrate that
# disappears from the pruned solves; DMRT; active (a different cut for each azimuth mode and the coherent solve)
PRUNE_FIXTURES = ["iba_L8_n12_prune", "iba_L6_n16_prune_substrate", "dmrt_L7_n12_prune",
                  "dmrt_L6_n10_prune_over_bad_layer"]  # the last one: layers that cannot be diagonalised below the cut
PRUNE_ACTIVE_FIXTURES = ["iba_active_L6_n10_prune"]
# heterogeneous snowpacks: a list of emmodels (one per layer) over layers mixing the two microstructure models
MIXED_FIXTURES = ["mixed_L4_n16_passive", "mixed_L4_n12_active"]
# emmodels without a device implementation (evaluated on the host, SMRT_EM_HOST): the reference's rayleigh on
# independent spheres, passive and active, and prescribed_kskaeps on a homogeneous microstructure
# IBA with emmodel_options=dict(dense_snow_correction="auto"): layers above half ice on the inverted medium (air in ice)
DENSE_AUTO_FIXTURES = ["iba_dense_auto_L5_n12", "iba_dense_auto_shs_active_L3_n8"]
# wet snow: layers at the melting point holding liquid water (grains coated in water, Maxwell Garnett in a water host); the
# active one with a very wet layer above half "ice + water" under dense_snow_correction="auto"
WET_FIXTURES = ["iba_wet_L4_n12_passive", "iba_wet_L3_n10_active", "dmrt_wet_L3_n12_passive"]
# IBA over four microstructure models, one per layer: teubner_strey, independent_sphere, exponential, sticky_hard_spheres
# ... and over the models on the unified parameters (porod length, polydispersity): scaled exponential, Teubner-Strey on
# both sides of polydispersity 1, sticky hard spheres -- reparametrisations of the device's closed forms
MICRO_FIXTURES = ["iba_micro4_L4_n12_passive", "iba_micro4_L4_n10_active", "iba_unified4_L4_n12_passive",
                  "iba_unified4_L4_n10_active"]
# the other members of IBA's family (iba_original, iba_maxwell_garnett; one per layer mixed with plain IBA): IBA's phase
# matrix assembled on the device, the scalars from the emmodel object (SMRT_EM_IBA_HOST)
IBA_FAMILY_FIXTURES = ["iba_original_L3_n12_passive", "iba_mg_L3_n10_active", "iba_family_L3_n12_passive",
                       # the symmetrised strong-contrast expansion: IBA's phase function at a complex wavenumber (passive)
                       "symsce_L3_n12_passive", "symsce_unified_L3_n12_passive", "symsce_ts_L3_n12_passive"]
# ... on the sphere models (sines of a complex argument: the dense route in the product; the oracle restates it)
SCE_SPHERES_FIXTURES = ["symsce_spheres_L3_n12_passive"]
HOST_EMMODEL_FIXTURES = ["rayleigh_L3_n16_passive", "rayleigh_L3_n12_active", "prescribed_L3_n16_passive"]
# ... together with process_coherent_layers: the phase matrices of the layers that stay live on the streams of the reduced
# snowpack (a 3 mm and a 6 mm layer leave at these frequencies)
COHERENT_HOST_FIXTURES = ["rayleigh_coherent_L5_n12_passive", "rayleigh_coherent_L5_n10_active"]


# DORT option process_coherent_layers: a 2 mm crust and a 3 mm ice lens become coherent interfaces, frequency by frequency
COHERENT_FIXTURES = ["coherent_L5_n16_passive", "coherent_L5_n12_active"]
# rough substrates in active mode: the dense reflection matrices of the bottom boundary come with the fixture (evaluated by
# the reference's geometrical_optics / iem_fung92 substrates) and are handed to the solver as numbers
ROUGH_SUBSTRATE_FIXTURES = ["rough_go_substrate_L2_n12_active", "rough_iem_substrate_L3_n10_active"]
# ... and the ones the reference runs in passive mode (diagonal in the streams: backscatter-only diffuse parts)
ROUGH_SUBSTRATE_PASSIVE_FIXTURES = ["rough_iem_substrate_L3_n10_passive", "rough_gob_substrate_L2_n12_passive"]


ROUGH_INTERFACE_FIXTURES = ["rough_iem_surface_L3_n10_passive", "rough_iem_inner_L3_n10_passive",
                            "rough_go_surface_L3_n10_active", "rough_iem_inner_L3_n10_active"]


def fixture_interfaces(d):
    """{i: {"Rtop": [per mode], "Ttop": ..., "Rbot": ..., "Tbot": ..., + "_coh"}} of a rough-interface fixture: the dense
    matrices of the interface on top of layer i as the reference combined them (inputs of the fixture)."""
    out = {}
    nm = (int(d["opt_m_max"]) + 1) if str(d["mode"]) == "A" else 1
    for i in np.atleast_1d(d["rough_interface"]):
        i = int(i)
        out[i] = {kind + tag: [d["itf%d_%s%s_m%d" % (i, kind, tag, m)] for m in range(nm)]
                  for kind in ("Rtop", "Ttop", "Rbot", "Tbot") for tag in ("", "_coh")}
    return out


class ReplayInterface:
    """An interface object speaking the reference's protocol (smrt/core/interface.py) that replays what the reference's
    own iem_fung92 / geometrical_optics object returned on the stream grids of a rough-interface fixture (itf_raw_*): lets
    the host-side evaluation (DORT.interface_matrices, Model.run) be tested without restating the interface physics."""

    def __init__(self, d):
        self.d = d
        self.eps_low = complex(d["itf_eps_low"][0])
        if "itf_raw_drf_up" in d:
            self.ft_even_diffuse_reflection_matrix = self._drf
        if "itf_raw_dtr_up" in d:
            self.ft_even_diffuse_transmission_matrix = self._dtr

    def _side(self, eps_1):
        return "up" if abs(complex(eps_1) - self.eps_low) < 1e-9 * abs(self.eps_low) else "dn"

    def _get(self, key, n_last):
        v = self.d["itf_raw_" + key]
        assert v.ndim == 0 or v.shape[-1] == n_last, (key, v.shape, n_last)
        return v

    def specular_reflection_matrix(self, frequency, eps_1, eps_2, mu1, npol):
        return self._get("spec_" + self._side(eps_1), len(mu1))

    def coherent_transmission_matrix(self, frequency, eps_1, eps_2, mu1, npol):
        return self._get("ctr_" + self._side(eps_1), len(mu1))

    def _drf(self, frequency, eps_1, eps_2, mu_s, mu_i, m_max, npol):
        return self._get("drf_" + self._side(eps_1), len(mu_i))

    def _dtr(self, frequency, eps_1, eps_2, mu_s, mu_i, m_max, npol):
        return self._get("dtr_" + self._side(eps_1), len(mu_i))


def pack_host_interfaces(per_pair, n_layers_max, n_max_stream, n_modes):
    """The (slot, matrices, coh) triple of PackedBatch(host_interfaces=...) from, per pair, a dict {i: dense dict} as
    fixture_interfaces returns it: zero-padded NE x NE blocks, transmissions cut to the common streams when they come as
    the square diagonal form (include/smrt_dort.h)."""
    ne = 3 * n_max_stream
    nslots = max(1, max(len(x) for x in per_pair))
    slot = -np.ones((len(per_pair), n_layers_max), np.int32)
    M = np.zeros((len(per_pair), nslots, n_modes, 4, ne, ne))
    coh = np.zeros((len(per_pair), nslots, 4, ne))
    for p, itfs in enumerate(per_pair):
        for k, (i, e) in enumerate(sorted(itfs.items())):
            slot[p, i] = k
            for m in range(n_modes):
                P = 2 if m == 0 else 3
                n_low, n_up = e["Rtop"][m].shape[0], e["Rbot"][m].shape[0]
                for q, (kind, rows) in enumerate((("Rtop", n_low), ("Ttop", n_up), ("Rbot", n_up), ("Tbot", n_low))):
                    A = np.asarray(e[kind][m], float)
                    r = min(A.shape[0], rows)
                    M[p, k, m, q, :r, :A.shape[1]] = A[:r]
                    if m == 0:
                        c = np.diag(np.asarray(e[kind + "_coh"][0], float))
                        nd = min(len(c), rows) if kind in ("Ttop", "Tbot") else len(c)
                        coh[p, k, q, :nd] = c[:nd]
    return slot, M, coh


def model_snowpack_from_fixture(d):
    """smrt_amd's own Snowpack object for a (uniform-microstructure) fixture, with the layer attributes the
    prescribed_kskaeps emmodel reads when the fixture holds them."""
    from smrt_amd import make_snowpack

    ms = str(d["microstructure"])
    kw = {k: d[k] for k in ("corr_length", "radius", "stickiness") if k in d}
    sp = make_snowpack(d["thickness"], ms, density=d["density"], temperature=d["temperature"], **kw)
    if "ks" in d:
        for l, lay in enumerate(sp.layers):
            lay.ks, lay.ka = float(d["ks"][l]), float(d["ka"][l])
            lay.effective_permittivity = complex(d["eps_re"][l], d["eps_im"][l])
    return sp


def host_batch_from_fixture(d):
    """The PackedBatch of a fixture whose emmodel has no device implementation: the product's own host evaluation
    (smrt_amd.rtsolver.dort.DORT._evaluate_on_host via _pack) -- runs without a GPU, only the launch needs one."""
    from smrt_amd.core.plugin import import_class
    from smrt_amd.core.sensor import active, passive
    from smrt_amd.rtsolver.dort import DORT

    o = fixture_options(d)
    solver = DORT(n_max_stream=o["n_max_stream"], m_max=o["m_max"], process_coherent_layers=fixture_coherent(d))
    sp = model_snowpack_from_fixture(d)
    act = str(d["mode"]) == "A"
    f0 = float(d["frequency"][0])
    sensor0 = active(f0, d["theta_inc_deg"]) if act else passive(f0, d["theta_deg"])
    base = import_class("emmodel", str(d["emmodel"]))

    class OwnPhase(base):   # a phase function "of its own" (same numbers): outside the Rayleigh family, whose members hand over
        def ft_even_phase(self, mu_s, mu_i, m_max, npol=None):   # their scalars only (SMRT_EM_RAYLEIGH_HOST) -- this helper
            return super().ft_even_phase(mu_s, mu_i, m_max, npol)  # is about the dense route (SMRT_EM_HOST)

    return solver._pack(sensor0, [sp], np.asarray(d["frequency"], float), [[(OwnPhase, {})] * sp.nlayer])


def fixture_emmodel(d):
    """The emmodel of a fixture in the oracle's form: a name, or a list of names (one per layer)."""
    return str(d["emmodel"]) if np.ndim(d["emmodel"]) == 0 else [str(e) for e in d["emmodel"]]
# configs[3] at full size: IBA, sentinel1(), 30 layers, 128 streams, m_max = 2 (N = 256 / 384 rows per azimuth mode)
BIG_ACTIVE_FIXTURES = ["cfg4_iba_active_L30_n128_sp0", "cfg4_iba_active_L30_n128_sp1"]
SIGMA_RTOL = 1e-8  # backscatter, relative (BASELINE.json north_star)


METHOD_RESULT_KEYS = ("result_eig", "result_half_rank_eig", "result_rerun")


def reference_method_spread(d):
    """Element-wise spread of the reference's own answers over its diagonalisation methods (stored in the active
    fixtures by tests/golden/add_method_spread.py as result_<method>); zeros when the fixture holds none."""
    ref = np.asarray(d["result"])
    spread = np.zeros_like(ref)
    for k in METHOD_RESULT_KEYS:   # (NOT every "result_*" key: result_incoherent of the coherent fixtures is another physics)
        if k in d:
            spread = np.maximum(spread, np.abs(np.asarray(d[k]) - ref))
    return spread


def oracle_method_spread(sp, frequency, theta_deg, ref, methods=("eig", "half_rank_eig"), **solve_kwargs):
    """The same spread for comparisons against the CPU oracle: |oracle(method) - ref| over the oracle's other
    diagonalisation methods (a method that fails on the case -- complex pairs on degenerate Rayleigh modes -- is
    skipped, like in the reference)."""
    from oracle import dort_oracle as O

    spread = np.zeros_like(np.asarray(ref))
    for method in methods:
        try:
            alt = O.solve(sp, frequency, theta_deg, method=method, **solve_kwargs)
        except O.OracleError:
            continue
        if np.all(np.isfinite(alt)):
            spread = np.maximum(spread, np.abs(alt - ref))
    return spread


# ---- audit of the backscatter bar: every comparison is recorded -- test, element kind, worst achieved
# relative error, what was allowed there, whether the plain 1e-8 bar had to be widened -- and written at session end to
# gpurun_out/parity_audit.txt (GPU box: the driver pulls it; the round's table is committed under profiles/).
# Widening is capped: beyond WIDEN_CAP a comparison fails unless the test is listed in WIDEN_WHITELIST with the spread the
# reference's own eigensolvers show on that fixture (measured, tests/golden/add_method_spread.py).
PARITY_AUDIT = []
WIDEN_CAP = 1e-6
WIDEN_WHITELIST = {
    # cross-polarised backscatter of this fixture is a cancellation to ~1e-3 of the modes: the reference's eig / half_rank_eig /
    # schur answers differ by 7.4e-4 relative there (HV / VH effectively unchecked, co-pol at the plain bar)
    "iba_shs_active_substrate_conditioning": 3e-3,
    "test_active_substrate_dominated_pair": 3e-3,   # (the GPU test on that fixture)
}


def _audit_context():
    cur = os.environ.get("PYTEST_CURRENT_TEST", "?")
    return cur.split(" (")[0]


def pytest_sessionfinish(session, exitstatus):
    if not PARITY_AUDIT:
        return
    out = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        rows = sorted(PARITY_AUDIT, key=lambda r: -r[2])
        nw = sum(1 for r in rows if r[4])
        with open(os.path.join(out, "parity_audit.txt"), "w") as fh:
            fh.write("# backscatter comparisons of this pytest session: %d element kinds, %d of them beyond the plain 1e-8 bar "
                     "(widened to 3 x the spread of the reference's own eigensolvers, capped at %.0e unless whitelisted)\n"
                     % (len(rows), nw, WIDEN_CAP))
            fh.write("# %-100s %-6s %10s %10s %s\n" % ("test", "kind", "achieved", "allowed", "widened"))
            for t, kind, ach, allowed, widened in rows:
                fh.write("%-102s %-6s %10.2e %10.2e %s\n" % (t, kind, ach, allowed, "WIDENED" if widened else "-"))
    except OSError:
        pass


def assert_backscatter_close(r, ref, rtol=SIGMA_RTOL, cross_rtol=SIGMA_RTOL, spread=None, spread_factor=3.0):
    """r, ref: [..., pol, pol_inc, theta_inc].  EVERY V,H x V,H intensity to 1e-8 relative on its own scale -- the
    north_star's bar -- co- and cross-polarised alike.  `spread` (same shape as ref, from reference_method_spread /
    oracle_method_spread) widens that to `spread_factor` times the disagreement among the reference's own eigensolvers
    where it is larger.  The disagreement is rounding noise, different at every angle, so it is pooled: the largest
    relative spread over the angles and over the two elements of a kind (HV | VH, or VV | HH) of one solve is the
    yardstick for all of them.  Background: the cross-polarised terms sit 30-50 dB below co-pol and come out of a
    cancellation between the azimuth modes, so the reference's answer is itself only defined to that spread (1e-12 ..
    1e-7 on the fixtures; measured on the GPU the device's deviation follows it pair by pair,
    profiles/r2_crosspol_probe.txt)."""
    r, ref = np.asarray(r), np.asarray(ref)
    rel = np.abs(r - ref)[..., :2, :2, :] / np.abs(ref[..., :2, :2, :])
    tol = np.empty_like(rel)
    tol[...] = rtol
    tol[..., 0, 1, :] = tol[..., 1, 0, :] = cross_rtol
    if spread is not None:
        srel = np.asarray(spread)[..., :2, :2, :] / np.abs(ref[..., :2, :2, :])
        co = np.maximum(srel[..., 0, 0, :], srel[..., 1, 1, :]).max(axis=-1, keepdims=True)      # pooled per solve
        cross = np.maximum(srel[..., 0, 1, :], srel[..., 1, 0, :]).max(axis=-1, keepdims=True)
        tol[..., 0, 0, :] = np.maximum(tol[..., 0, 0, :], spread_factor * co)
        tol[..., 1, 1, :] = np.maximum(tol[..., 1, 1, :], spread_factor * co)
        tol[..., 0, 1, :] = np.maximum(tol[..., 0, 1, :], spread_factor * cross)
        tol[..., 1, 0, :] = np.maximum(tol[..., 1, 0, :], spread_factor * cross)
    ctx_name = _audit_context()
    cap = max([WIDEN_CAP] + [v for k, v in WIDEN_WHITELIST.items() if k in ctx_name])
    plain = np.empty_like(rel)
    plain[...] = rtol
    plain[..., 0, 1, :] = plain[..., 1, 0, :] = cross_rtol
    assert (tol <= np.maximum(plain, cap)).all(), (
        "the spread of the reference's own methods would widen the backscatter bar to %.1e (> cap %.0e): add the fixture to "
        "conftest.WIDEN_WHITELIST with its measured spread if that is what the reference does" % (tol.max(), cap))
    for kind, idx in (("co", [(0, 0), (1, 1)]), ("cross", [(0, 1), (1, 0)])):
        a = max(float(rel[..., i, j, :].max()) for i, j in idx)
        t_ = max(float(tol[..., i, j, :].max()) for i, j in idx)
        p_ = max(float(plain[..., i, j, :].max()) for i, j in idx)
        PARITY_AUDIT.append((ctx_name, kind, a, t_, bool(t_ > p_ and a > p_)))
    bad = rel > tol
    assert not bad.any(), "backscatter off by up to %.2e relative (allowed %.2e there)" % (rel[bad].max(), tol[bad].max())
    # third Stokes rows/columns are multiplied by sin(m pi) ~ 1e-16 in backscatter: only their level is meaningful
    assert np.abs(r[..., 2, :, :]).max() <= 10 * np.abs(ref[..., 2, :, :]).max() + 1e-30


# ---- stand-ins for the reference's objects (tests/golden/reference_objects.json, made by make_binding_dump.py) --------
def _undump(v):
    if isinstance(v, dict) and "complex" in v:
        return complex(*v["complex"])
    if isinstance(v, dict) and "dict" in v:
        return {(tuple(k) if isinstance(k, list) else k): _undump(x) for k, x in v["dict"]}
    if isinstance(v, list):
        return [_undump(x) for x in v]
    return v


_STANDIN_CLASSES = {}


def standin_class(module, name, **members):
    """A class that carries the reference class's identity (module, name) and nothing of its code."""
    key = (module, name)
    if key not in _STANDIN_CLASSES:
        _STANDIN_CLASSES[key] = type(name, (), dict(__module__=module, **members))
    return _STANDIN_CLASSES[key]


def standin_function(module, name):
    """A function that carries the identity (module, name) of one of the reference's permittivity functions."""
    def f(*a, **k):
        raise AssertionError("a stand-in permittivity function is never evaluated")
    f.__module__, f.__name__ = module, name
    return f


def _standin(cls_id, **attrs):
    obj = standin_class(*cls_id)()
    for k, v in attrs.items():
        setattr(obj, k, v)
    return obj


def standins_from_dump(case):
    """(model, simulations, snowpacks, expected results) of one case of reference_objects.json: objects with the classes'
    names, modules and public attributes of the reference's -- what its `Model.run` would hand to a runner -- and none of
    its behaviour beyond `substrate.permittivity(frequency)` (a table) and a constructor for the emmodel class."""
    function = standin_function

    def layer(d):
        ms = _standin(d["microstructure"]["cls"], **{k: _undump(v) for k, v in d["microstructure"]["attrs"].items()})
        pm = tuple(function(*p["function"]) if isinstance(p, dict) and "function" in p else _undump(p) for p in d["permittivity_model"])
        return _standin(d["cls"], microstructure=ms, microstructure_model=type(ms), permittivity_model=pm,
                        frac_volume=d["frac_volume"], **{k: _undump(v) for k, v in d["attrs"].items()})

    def substrate(d):
        if d is None:
            return None
        table = None if d["permittivity"] is None else {f: _undump(e) for f, e in d["permittivity"]}
        cls = standin_class(*d["cls"], permittivity=lambda self, frequency: self._table[frequency],
                            specular_reflection_matrix=lambda self, *a: None)
        obj = cls()
        obj.temperature, obj._table = d["temperature"], table
        obj.specular_reflection = _undump(d["specular_reflection"])
        return obj

    def snowpack(d):
        layers = [layer(x) for x in d["layers"]]
        sp = _standin(d["cls"], layers=layers, interfaces=[standin_class(*c)() for c in d["interfaces"]],
                      substrate=substrate(d["substrate"]),
                      atmosphere=None if d["atmosphere"] is None else
                      _standin(d["atmosphere"]["cls"], **{k: _undump(v) for k, v in d["atmosphere"]["attrs"].items()}))
        type(sp).nlayer = property(lambda self: len(self.layers))
        type(sp).layer_thicknesses = property(lambda self: [lay.thickness for lay in self.layers])
        return sp

    def sensor(d):
        attrs = {k: _undump(v) for k, v in d["attrs"].items()}
        for k in ("theta_deg", "theta", "theta_inc_deg", "theta_inc"):
            attrs[k] = None if attrs[k] is None else np.asarray(attrs[k], float)
        return _standin(d["cls"], mode=d["mode"], **attrs)

    def emmodel_init(self, sensor, layer, **options):
        self.options = options

    md = case["model"]
    emmodel = standin_class(*md["emmodel"], __init__=emmodel_init)
    model_cls = standin_class(*md["cls"], run_single_simulation=lambda self, simulation, atmosphere, parallel_computation: None)
    model = model_cls()
    from smrt_amd.rtsolver.dort import DORT

    model.emmodel, model.emmodel_options = emmodel, _undump(md["emmodel_options"])
    model.rtsolver, model.rtsolver_options = DORT, _undump(md["rtsolver_options"])
    sensors = [sensor(s) for s in case["sensors"]]
    packs = [snowpack(s) for s in case["snowpacks"]]
    sims = [(sensors[i], packs[j]) for i, j in case["simulations"]]
    return model, sims, packs, [np.asarray(r) for r in case["results"]]


def load_reference_objects():
    import json

    with open(os.path.join(GOLDEN, "reference_objects.json")) as f:
        return {c["name"]: c for c in json.load(f)["cases"]}


# what the rough-interface / rough-substrate fixtures were made with (tests/golden/make_golden.py): model name + parameters
ROUGH_INTERFACE_MODELS = {
    "rough_iem_surface_L3_n10_passive": ("iem_fung92", dict(roughness_rms=0.002, corr_length=0.05)),
    "rough_iem_inner_L3_n10_passive": ("iem_fung92", dict(roughness_rms=0.002, corr_length=0.05)),
    "rough_go_surface_L3_n10_active": ("geometrical_optics", dict(mean_square_slope=0.03)),
    "rough_iem_inner_L3_n10_active": ("iem_fung92", dict(roughness_rms=0.002, corr_length=0.05)),
}
ROUGH_SUBSTRATE_MODELS = {
    "rough_go_substrate_L2_n12_active": ("geometrical_optics", dict(mean_square_slope=0.05)),
    "rough_iem_substrate_L3_n10_active": ("iem_fung92", dict(roughness_rms=0.004, corr_length=0.05)),
    "rough_iem_substrate_L3_n10_passive": ("iem_fung92", dict(roughness_rms=0.004, corr_length=0.05)),
    "rough_gob_substrate_L2_n12_passive": ("geometrical_optics_backscatter", dict(mean_square_slope=0.05)),
}


# DORT options x rough interfaces / substrates: the inputs of tests/golden/make_rough_option_fixtures.py (the fixtures of
# these names hold the reference's results for them)
_OPT_LAYERS = dict(density=[250.0, 320.0, 380.0, 400.0], temperature=[258.0, 261.0, 264.0, 266.0],
                   corr_length=[1e-4, 3e-4, 2e-4, 1.5e-4])
ROUGH_OPTION_CASES = {
    "rough_prune_iem_L4_n10_passive": dict(
        thickness=[0.3, 2.0, 0.5, 100.0], **_OPT_LAYERS, mode="P", frequency=36.5e9, theta=[40.0, 55.0],
        options=dict(n_max_stream=10, prune_deep_snowpack=0.5),
        interface=("iem_fung92", dict(roughness_rms=0.002, corr_length=0.05), 2)),
    "rough_prune_go_L4_n10_active": dict(
        thickness=[0.3, 2.0, 0.5, 1000.0], **_OPT_LAYERS, mode="A", frequency=36.5e9, theta=[30.0, 40.0],
        options=dict(n_max_stream=10, m_max=2, prune_deep_snowpack=0.5),
        interface=("geometrical_optics", dict(mean_square_slope=0.03), 2)),
    "rough_coherent_iem_L5_n10_passive": dict(
        thickness=[0.3, 0.25, 0.002, 0.4, 100.0], density=[250.0, 300.0, 900.0, 350.0, 400.0],
        temperature=[258.0, 260.0, 261.0, 263.0, 265.0], corr_length=[1e-4, 1.5e-4, 5e-5, 2e-4, 1.5e-4], mode="P",
        frequency=18.7e9, theta=[40.0, 55.0], options=dict(n_max_stream=10, process_coherent_layers=True),
        interface=("iem_fung92", dict(roughness_rms=0.002, corr_length=0.05), 0)),
    "rough_coherent_adjacent_L5_n10_passive": dict(
        thickness=[0.3, 0.25, 0.002, 0.4, 100.0], density=[250.0, 300.0, 900.0, 350.0, 400.0],
        temperature=[258.0, 260.0, 261.0, 263.0, 265.0], corr_length=[1e-4, 1.5e-4, 5e-5, 2e-4, 1.5e-4], mode="P",
        frequency=18.7e9, theta=[40.0, 55.0], options=dict(n_max_stream=10, process_coherent_layers=True),
        interface=("iem_fung92", dict(roughness_rms=0.002, corr_length=0.05), 2)),
    "rough_coherent_gosub_L4_n10_active": dict(
        thickness=[0.3, 0.002, 0.4, 0.8], density=[250.0, 900.0, 350.0, 400.0], temperature=[258.0, 261.0, 263.0, 265.0],
        corr_length=[1e-4, 5e-5, 2e-4, 1.5e-4], mode="A", frequency=13.4e9, theta=[30.0, 40.0],
        options=dict(n_max_stream=10, m_max=2, process_coherent_layers=True),
        substrate=("geometrical_optics", dict(mean_square_slope=0.05)), substrate_eps=(8.0, 1.0), substrate_temperature=268.0),
}


# ---- the device source on the CPU, behind the product's own host code --------------------------------------------------
EMU_LIB = os.path.join(ROOT, "tests", "hostemu", "libsmrt_emu.so")


class EmulatedContext:
    """Stands where rtsolver/dort.py:get_context returns the GPU context: same `run` signature, the device source run by
    the CPU emulator; records every batch it is handed."""

    def __init__(self):
        if not os.path.exists(EMU_LIB):
            pytest.skip("emulator library not built (python __graft_entry__.py)")
        import ctypes as C
        import threading

        from smrt_amd._native import SmrtBatch

        self.lib = C.CDLL(EMU_LIB)
        P = C.POINTER
        self.lib.smrt_emu_run.argtypes = [P(SmrtBatch), C.c_longlong, C.c_longlong, C.c_int, C.c_int, P(C.c_double),
                                          P(C.c_int32), P(C.c_double), P(C.c_double), P(C.c_double), P(C.c_long)]
        self.lock = threading.RLock()
        self.batches = []
        self.compute = True    # False: record the batch, hand back zeros (20 layers x 32 streams cost ~9 s per pair here)

    def set_block_threads(self, n):
        pass

    def run(self, batch, lo=0, n=None, pairs=None):
        import ctypes as C

        from smrt_amd._native import BatchOutput

        self.batches.append((batch, None if pairs is None else np.array(pairs)))
        todo = [(int(lo), batch.n_pairs - int(lo) if n is None else int(n))] if pairs is None else [(int(p), 1) for p in pairs]
        out = BatchOutput(batch, sum(c for _, c in todo))
        if not self.compute:
            for a in (out.values, out.status, out.layers, out.streams):
                a[...] = 0
            return out
        dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))  # noqa: E731
        row = 0
        for begin, count in todo:
            sl = slice(row, row + count)
            v, st = np.empty_like(out.values[sl]), np.empty(count, np.int32)
            lay, stream = np.empty_like(out.layers[sl]), np.empty_like(out.streams[sl])
            rc = self.lib.smrt_emu_run(C.byref(batch.struct), begin, count, 64, 0, dp(v), st.ctypes.data_as(C.POINTER(C.c_int32)),
                                       dp(lay), dp(stream), None, None)
            assert rc == 0
            out.values[sl], out.status[sl], out.layers[sl], out.streams[sl] = v, st, lay, stream
            row += count
        return out




@pytest.fixture()
def emulated(monkeypatch):
    """rtsolver/dort.py:get_context routed to the CPU emulator of the device source (EmulatedContext): Model.run, the
    runners and the host-side packing run as they are, the kernels run under tests/hostemu."""
    import smrt_amd.rtsolver.dort as dort

    ctx = EmulatedContext()
    monkeypatch.setattr(dort, "get_context", lambda device=None: ctx)
    return ctx
