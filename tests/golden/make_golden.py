#!/usr/bin/env python
"""Golden-vector generator.  RUNS ONLY IN THE BUILD CONTAINER (needs /root/reference).

It imports the reference package (pure Python) from /root/reference with two tiny stand-in modules for the
absent third-party packages `xarray` and `numba` (tests/golden/_refstubs), drives the reference exactly the way
`Model.run` drives it (prepare_simulations -> run_single_simulation, smrt/core/model.py:415-619), and stores
inputs + outputs (+ a few intermediate stages) as small .npz fixtures next to this script.

Nothing here travels as code to the GPU box: the fixtures are data (inputs and expected outputs).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py
"""
import os
import sys
import warnings

HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(HERE, "_refstubs"))
sys.path.insert(0, "/root/reference")

import numpy as np  # noqa: E402

warnings.filterwarnings("ignore")

from smrt import make_model, make_snowpack, sensor_list  # noqa: E402
from smrt.core.sensor import active, passive  # noqa: E402
from smrt.rtsolver.dort import DORT  # noqa: E402
from smrt.rtsolver.rtsolver_utils import compute_interface_properties  # noqa: E402


def snowpack_arrays(sp):
    """Flatten a reference Snowpack into plain arrays (the layout our own boundary consumes)."""
    L = len(sp.layers)
    out = dict(
        thickness=np.array([lay.thickness for lay in sp.layers], float),
        density=np.array([lay.density for lay in sp.layers], float),
        temperature=np.array([lay.temperature for lay in sp.layers], float),
        frac_volume=np.array([lay.frac_volume for lay in sp.layers], float),
    )
    lw = [float(getattr(lay, "liquid_water", 0) or 0) for lay in sp.layers]
    if any(lw):   # wet snow: water / (ice + water) volume per layer (frac_volume is then ice + water)
        out["liquid_water"] = np.array(lw)
    kinds = {type(lay.microstructure).__name__ for lay in sp.layers}
    if len(kinds) > 1:   # mixed microstructure models: one name per layer, NaN for the parameters a layer does not have
        get = lambda lay, a: float(getattr(lay.microstructure, a, np.nan))  # noqa: E731
        names = {"Exponential": "exponential", "StickyHardSpheres": "sticky_hard_spheres",
                 "IndependentSphere": "independent_sphere", "TeubnerStrey": "teubner_strey",
                 "UnifiedScaledExponential": "unified_scaled_exponential", "UnifiedTeubnerStrey": "unified_teubner_strey",
                 "UnifiedStickyHardSpheres": "unified_sticky_hard_spheres"}
        out["microstructure"] = np.array([names[type(lay.microstructure).__name__] for lay in sp.layers])
        # (the unified models derive corr_length / radius internally: only their own parameters are inputs -- NaN elsewhere)
        for a in ("corr_length", "radius", "stickiness", "repeat_distance", "porod_length", "polydispersity"):
            vals = [np.nan if (a in ("corr_length", "radius", "stickiness", "repeat_distance") and
                               type(lay.microstructure).__name__.startswith("Unified")) else get(lay, a) for lay in sp.layers]
            if not np.all(np.isnan(vals)):
                out[a] = np.array(vals)
        return out
    ms = sp.layers[0].microstructure
    if getattr(sp.layers[0], "ks", None) is not None:   # prescribed_kskaeps reads these layer attributes
        out["ks"] = np.array([lay.ks for lay in sp.layers], float)
        out["ka"] = np.array([lay.ka for lay in sp.layers], float)
        out["eps_re"] = np.array([complex(lay.effective_permittivity).real for lay in sp.layers])
        out["eps_im"] = np.array([complex(lay.effective_permittivity).imag for lay in sp.layers])
    if type(ms).__name__ == "IndependentSphere":
        out["microstructure"] = "independent_sphere"
        out["radius"] = np.array([lay.microstructure.radius for lay in sp.layers], float)
    elif type(ms).__name__ == "Homogeneous":
        out["microstructure"] = "homogeneous"
    elif type(ms).__name__ in ("UnifiedScaledExponential", "UnifiedTeubnerStrey"):
        out["microstructure"] = {"UnifiedScaledExponential": "unified_scaled_exponential", "UnifiedTeubnerStrey": "unified_teubner_strey"}[type(ms).__name__]
        out["porod_length"] = np.array([lay.microstructure.porod_length for lay in sp.layers], float)
        out["polydispersity"] = np.array([lay.microstructure.polydispersity for lay in sp.layers], float)
    elif hasattr(ms, "corr_length"):
        out["microstructure"] = "exponential"
        out["corr_length"] = np.array([lay.microstructure.corr_length for lay in sp.layers], float)
    elif hasattr(ms, "stickiness"):
        out["microstructure"] = "sticky_hard_spheres"
        out["radius"] = np.array([lay.microstructure.radius for lay in sp.layers], float)
        out["stickiness"] = np.array([lay.microstructure.stickiness for lay in sp.layers], float)
    else:
        raise RuntimeError("unexpected microstructure")
    assert L == len(out["thickness"])
    return out


def run_case(emmodel, sensor, sp, rtsolver_options=None, stages=False, stage_layers=(0,), emmodel_options=None,
             emmodel_label=None):
    """Run the reference for every (frequency) configuration of `sensor` on snowpack `sp`."""
    if ONLY and SKIP_OLD[0]:
        return {}
    rtsolver_options = dict(rtsolver_options or {})
    # emmodel: a name, or a list (one per layer)
    m = make_model(emmodel, "dort", rtsolver_options=rtsolver_options, emmodel_options=emmodel_options or {})
    if emmodel_label is not None:   # the name under which the oracle / the tests know (emmodel, emmodel_options)
        emmodel = emmodel_label
    sims, dims = m.prepare_simulations(sensor, sp, None, "snowpack")
    sims = list(sims)
    out = dict(snowpack_arrays(sp))
    out["emmodel"] = np.array(emmodel) if isinstance(emmodel, (list, tuple)) else emmodel
    out["mode"] = sensor.mode
    freqs, datas = [], []
    for isim, sim in enumerate(sims):
        se, spk = sim
        r = m.run_single_simulation(sim, None, None)
        freqs.append(float(se.frequency))
        datas.append(np.asarray(r.data.values))
        tag = "f%d_" % isim
        for k in ("stream_angles", "effective_permittivity", "ks", "ke", "ka"):
            out[tag + k] = np.asarray(r.other_data[k].values)
        if stages:
            dump_stages(out, tag, m, se, spk, rtsolver_options, stage_layers)
    out["frequency"] = np.array(freqs)
    out["result"] = np.array(datas)  # (nfreq, pol, theta) or (nfreq, pol_inc, pol, theta_inc)
    out["theta_deg"] = np.asarray(sensor.theta_deg, float)
    if sensor.mode == "A":
        out["theta_inc_deg"] = np.asarray(sensor.theta_inc_deg, float)
    for k, v in rtsolver_options.items():
        out["opt_" + k] = np.asarray(v)
    sub = sp.substrate
    if sub is not None:  # what our boundary needs of it: kind, permittivity or reflection per frequency, temperature
        kind = type(sub).__name__.lower()
        out["substrate_kind"] = "flat" if kind == "flat" else "reflector"
        out["substrate_temperature"] = np.nan if sub.temperature is None else float(sub.temperature)
        if kind == "flat":
            out["substrate_eps"] = np.array([complex(sub.permittivity(f)) for f in freqs])
        else:
            one = np.array([1.0])
            out["substrate_R"] = np.array([[float(sub._get_refl(f, pol, one)[0]) for pol in "VH"] for f in freqs])
    atm = sp.atmosphere
    if atm is not None:
        pick = lambda x, f: float(x[f]) if isinstance(x, dict) else float(x)  # noqa: E731
        out["atm_tb_down"] = np.array([pick(atm.constant_tbdown, f) for f in freqs])
        out["atm_tb_up"] = np.array([pick(atm.constant_tbup, f) for f in freqs])
        out["atm_trans"] = np.array([pick(atm.constant_trans, f) for f in freqs])
    return out


def dump_stages(out, tag, m, se, spk, rtsolver_options, stage_layers):
    """Intermediate tensors of one solve: streams, interface diagonals, A matrices, sorted eigenvalues."""
    ems = m.prepare_emmodels(se, spk)
    solver = DORT(**rtsolver_options)
    solver.init_solve(spk, ems, se, None)
    solver.prepare_streams()
    st = solver.streams
    L = len(ems)
    nmax = max(st.n)
    mu = np.zeros((L, nmax))
    w = np.zeros((L, nmax))
    for l in range(L):
        mu[l, : st.n[l]] = st.mu[l]
        w[l, : st.n[l]] = st.weight[l]
    out[tag + "streams_n"] = np.asarray(st.n, int)
    out[tag + "streams_mu"] = mu
    out[tag + "streams_weight"] = w
    out[tag + "streams_outmu"] = np.asarray(st.outmu)
    out[tag + "streams_outweight"] = np.asarray(st.outweight)
    npol = 2 if se.mode == "P" else 3
    m_max = solver.m_max if se.mode == "A" else 0
    itf = compute_interface_properties(
        se.frequency, spk.interfaces, spk.substrate, solver.effective_permittivity, st, m_max, npol
    )
    for name, fn in (
        ("Rtop", itf.reflection_top),
        ("Ttop", itf.transmission_top),
        ("Rbottom", itf.reflection_bottom),
        ("Tbottom", itf.transmission_bottom),
    ):
        for mode in range(m_max + 1):
            P = 2 if mode == 0 else 3
            arr = np.zeros((L + 1, nmax * P))
            for l in list(range(L)) + [-1]:
                if l == -1 and name in ("Rtop", "Ttop"):
                    continue
                d = fn(l, mode, False)
                d = np.zeros(0) if np.isscalar(d) or getattr(d, "shape", None) == () else np.asarray(d.diagonal())
                arr[l if l >= 0 else L, : len(d)] = d  # row L holds the air->snow side (index -1)
            out[tag + "itf_%s_m%d" % (name, mode)] = arr
    from smrt.rtsolver.dort import EigenValueSolver

    for mode in range(m_max + 1):
        P = 2 if mode == 0 else 3
        betas = np.zeros((L, 2 * nmax * P))
        for l in range(L):
            es = EigenValueSolver(
                ke=ems[l].ke,
                ks=ems[l].ks,
                ft_even_phase_function=ems[l].ft_even_phase,
                mu=st.mu[l],
                weight=st.weight[l],
                m_max=m_max,
                method="schur_forcedtriu",
                normalization=getattr(ems[l], "_respect_reciprocity_principle", True),
                symmetrization=False,
                cache=False,
            )
            if mode > 0:
                es.solve(0, False)  # normalisation of mode m needs mode 0 first (dort.py:803-807)
            if l in stage_layers:
                A = es.solve_generic(mode, False, debug_A=True)
                out[tag + "A_m%d_l%d" % (mode, l)] = np.asarray(A)
            beta, Eu, Ed = es.solve(mode, False)
            betas[l, : len(beta)] = np.sort(beta)
        out[tag + "beta_sorted_m%d" % mode] = betas


ONLY = set(sys.argv[1:])  # optional: regenerate only the named fixtures
SKIP_OLD = [False]  # set while main() walks through cases that are not wanted (their run is skipped)


def wanted(name):
    return not ONLY or name in ONLY


def save(name, d):
    if not wanted(name) or not d:
        return
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in d.items()})
    print("wrote", os.path.relpath(path), "%.1f KB" % (os.path.getsize(path) / 1024))


def random_snowpack(rng, L, micro, thick_lo=0.05, thick_hi=0.30, last=100.0):
    """Synthetic snowpack laws of SURVEY.md section 8d."""
    thickness = np.append(rng.uniform(thick_lo, thick_hi, L - 1), last)
    density = rng.uniform(150, 450, L)
    temperature = rng.uniform(230, 270, L)
    if micro == "exponential":
        size = rng.uniform(5e-5, 3e-4, L)
        return make_snowpack(thickness, "exponential", density=density, temperature=temperature, corr_length=size)
    size = rng.uniform(5e-5, 1.5e-4, L)
    return make_snowpack(
        thickness, "sticky_hard_spheres", density=density, temperature=temperature, radius=size, stickiness=0.2
    )


def run_new(*args, **kwargs):
    """run_case for the cases that are guarded by wanted(): always runs."""
    SKIP_OLD[0] = False
    try:
        return run_case(*args, **kwargs)
    finally:
        SKIP_OLD[0] = bool(ONLY)


def main():
    SKIP_OLD[0] = bool(ONLY)  # with fixture names on the command line the unguarded (older) cases are not re-run
    # (i) config 1: examples/iba_onelayer_example.py:6-27
    sp = make_snowpack([100], "exponential", density=[320], temperature=[270], corr_length=[5e-5])
    d = run_case("iba", sensor_list.amsre("37V"), sp, stages=True)
    em = make_model("iba", "dort").prepare_emmodels(next(sensor_list.amsre("37V").iterate("frequency")) if False else
                                                    sensor_list.passive(36.5e9, 55), sp)[0]
    d["iba_coeff"] = em.iba_coeff
    save("cfg1_iba_onelayer", d)

    # (ii) smrt/test/test_integration_iba.py:13-69 (2-layer passive 37 GHz, active 19 GHz 55 deg)
    sp2 = make_snowpack(
        thickness=[0.1, 100], microstructure_model="exponential", density=[200, 400],
        temperature=[250.0, 250.0], corr_length=[5e-5, 5e-5],
    )
    save("iba_2layer_passive37", run_case("iba", sensor_list.amsre("37V"), sp2, stages=True, stage_layers=(0, 1)))
    save("iba_2layer_active19", run_case("iba", active(frequency=19e9, theta_inc=55), sp2, stages=True,
                                         stage_layers=(0,)))

    # (iii) smrt/test/test_dmrtdort.py:20-37 snowpack with dmrt_qca_shortrange
    sp3 = make_snowpack(
        [0.1, 1000], "sticky_hard_spheres", density=[200, 400], temperature=[250.0, 250.0],
        radius=[2e-4, 2e-4], stickiness=[0.1, 0.1],
    )
    save("dmrt_2layer_passive37", run_case("dmrt_qca_shortrange", sensor_list.amsre("37V"), sp3, stages=True,
                                           stage_layers=(0,)))

    # (iv) random multilayer cases
    rng = np.random.default_rng(2)
    sps = [random_snowpack(rng, 20, "exponential") for _ in range(3)]
    for i, spx in enumerate(sps[:2]):
        save("cfg2_iba_L20_n32_sp%d" % i, run_case("iba", sensor_list.amsre(), spx, stages=(i == 0),
                                                   stage_layers=(0, 19)))
    # reduced-size passive cases for fast oracle checks, several viewing angles incl. extrapolation to nadir
    rng = np.random.default_rng(12)
    spx = random_snowpack(rng, 6, "exponential")
    save("iba_L6_n8_angles", run_case("iba", passive([10.65e9, 36.5e9, 89e9], [0, 5, 30, 55, 70]), spx,
                                      rtsolver_options=dict(n_max_stream=8), stages=True, stage_layers=(0, 5)))
    spx = random_snowpack(rng, 3, "exponential", last=0.5)  # optically shallow, no substrate
    save("iba_L3_n16_shallow", run_case("iba", passive([18.7e9, 36.5e9], [40, 55]), spx,
                                        rtsolver_options=dict(n_max_stream=16), stages=True, stage_layers=(0,)))
    rng = np.random.default_rng(3)
    spx = random_snowpack(rng, 50, "sticky_hard_spheres")
    save("cfg3_dmrt_L50_n64_sp0", run_case("dmrt_qca_shortrange", passive([6.925e9, 36.5e9, 89e9], 55), spx,
                                           rtsolver_options=dict(n_max_stream=64)))
    spx = random_snowpack(rng, 8, "sticky_hard_spheres")
    save("dmrt_L8_n16", run_case("dmrt_qca_shortrange", passive([10.65e9, 89e9], [30, 55]), spx,
                                 rtsolver_options=dict(n_max_stream=16), stages=True, stage_layers=(0,)))
    # active, reduced size (cfg4 laws: thin layers, C band) and one at 32 streams
    rng = np.random.default_rng(4)
    spx = random_snowpack(rng, 5, "exponential", 0.02, 0.10, 1000.0)
    save("cfg4_iba_active_L5_n16", run_case("iba", sensor_list.sentinel1(), spx,
                                            rtsolver_options=dict(n_max_stream=16, m_max=2), stages=True,
                                            stage_layers=(0,)))
    spx = random_snowpack(rng, 4, "exponential", 0.02, 0.10, 1000.0)
    save("iba_active_L4_n32_ku", run_case("iba", active(13.4e9, [30, 40]), spx,
                                          rtsolver_options=dict(n_max_stream=32, m_max=2)))

    # active, other branches: DMRT (closed-form Rayleigh modes), sticky hard spheres under IBA, m_max = 1, and an
    # incidence angle steeper than every stream (mu = 1 node of the active interpolation)
    rng = np.random.default_rng(5)
    spx = random_snowpack(rng, 3, "sticky_hard_spheres", 0.05, 0.20, 1000.0)
    if wanted("dmrt_active_L3_n12"):
        save("dmrt_active_L3_n12", run_new("dmrt_qca_shortrange", active(10e9, [35, 50]), spx,
                                            rtsolver_options=dict(n_max_stream=12, m_max=2)))
    if wanted("iba_shs_active_L3_n8"):
        save("iba_shs_active_L3_n8", run_new("iba", active(17.2e9, [25, 45]), spx,
                                              rtsolver_options=dict(n_max_stream=8, m_max=2)))
    spx = random_snowpack(rng, 3, "exponential", 0.05, 0.20, 1000.0)
    if wanted("iba_active_L3_n10_m1_steep"):
        save("iba_active_L3_n10_m1_steep", run_new("iba", active(9.6e9, [2, 30, 55]), spx,
                                                    rtsolver_options=dict(n_max_stream=10, m_max=1)))

    # substrates (Flat with a constant permittivity, Reflector) and a SimpleIsotropicAtmosphere
    # (smrt/substrate/flat.py, reflector.py, atmosphere/simple_isotropic_atmosphere.py; physics of
    # smrt/test/test_physics_law.py)
    from smrt.atmosphere.simple_isotropic_atmosphere import SimpleIsotropicAtmosphere
    from smrt.substrate.flat import Flat
    from smrt.substrate.reflector import Reflector

    def snow(rng, L, micro, last, **kw):
        base = random_snowpack(rng, L, micro, 0.05, 0.30, last)
        return make_snowpack([lay.thickness for lay in base.layers], micro,
                             density=[lay.density for lay in base.layers],
                             temperature=[lay.temperature for lay in base.layers],
                             **({"corr_length": [lay.microstructure.corr_length for lay in base.layers]}
                                if micro == "exponential" else
                                {"radius": [lay.microstructure.radius for lay in base.layers], "stickiness": 0.2}), **kw)

    rng = np.random.default_rng(6)
    if wanted("iba_L3_n16_flat_substrate"):
        spx = snow(rng, 3, "exponential", 0.4, substrate=Flat(temperature=270.0, permittivity_model=3.5 + 0.3j))
        save("iba_L3_n16_flat_substrate", run_new("iba", passive([18.7e9, 36.5e9], [40, 55]), spx,
                                                   rtsolver_options=dict(n_max_stream=16)))
    if wanted("iba_L3_n16_substrate_atmosphere"):
        spx = snow(rng, 3, "exponential", 0.3, substrate=Flat(temperature=268.0, permittivity_model=6.0 + 0.8j),
                   atmosphere=SimpleIsotropicAtmosphere(tb_down={18.7e9: 20.0, 36.5e9: 32.0},
                                                        tb_up={18.7e9: 6.0, 36.5e9: 11.0},
                                                        transmittance={18.7e9: 0.95, 36.5e9: 0.9}))
        save("iba_L3_n16_substrate_atmosphere", run_new("iba", passive([18.7e9, 36.5e9], [30, 55]), spx,
                                                         rtsolver_options=dict(n_max_stream=16)))
    if wanted("dmrt_L4_n12_reflector"):
        spx = snow(rng, 4, "sticky_hard_spheres", 0.25,
                   substrate=Reflector(temperature=265.0, specular_reflection={"V": 0.2, "H": 0.35}))
        save("dmrt_L4_n12_reflector", run_new("dmrt_qca_shortrange", passive([10.65e9, 36.5e9], [55]), spx,
                                               rtsolver_options=dict(n_max_stream=12)))
    if wanted("iba_L2_n10_mirror_atmosphere_only"):
        spx = snow(rng, 2, "exponential", 0.2, substrate=Reflector(specular_reflection=1.0),
                   atmosphere=SimpleIsotropicAtmosphere(tb_down=25.0, tb_up=8.0, transmittance=0.92))
        save("iba_L2_n10_mirror_atmosphere_only", run_new("iba", passive([23.8e9], [0, 50]), spx,
                                                           rtsolver_options=dict(n_max_stream=10)))
    if wanted("iba_active_L3_n12_flat_substrate"):
        spx = snow(rng, 3, "exponential", 0.15, substrate=Flat(temperature=270.0, permittivity_model=4.0 + 0.5j))
        save("iba_active_L3_n12_flat_substrate", run_new("iba", active(13.4e9, [25, 45]), spx,
                                                          rtsolver_options=dict(n_max_stream=12, m_max=2)))

    # other emmodels feeding the same solver: DMRT QCA-CP short range (smrt/test/test_dmrtdort.py:20-37 known answer
    # 201.83572222 / 187.29558162 K) and the non-scattering medium over a substrate
    if wanted("dmrtcp_2layer_passive37"):
        save("dmrtcp_2layer_passive37", run_new("dmrt_qcacp_shortrange", sensor_list.amsre("37V"), sp3))
    rng = np.random.default_rng(7)
    if wanted("dmrtcp_L5_n12"):
        spx = random_snowpack(rng, 5, "sticky_hard_spheres")
        save("dmrtcp_L5_n12", run_new("dmrt_qcacp_shortrange", passive([10.65e9, 36.5e9], [35, 55]), spx,
                                      rtsolver_options=dict(n_max_stream=12)))
    if wanted("nonscattering_L3_n10_substrate"):
        spx = snow(rng, 3, "exponential", 0.5, substrate=Flat(temperature=271.0, permittivity_model=5.0 + 0.6j))
        save("nonscattering_L3_n10_substrate", run_new("nonscattering", passive([6.925e9, 36.5e9], [20, 55]), spx,
                                                       rtsolver_options=dict(n_max_stream=10)))

    # DORT option prune_deep_snowpack (dort.py:117-124,176-178,443-452): layers below the optical depth are dropped.
    # Chosen so that the pack is cut at different layers for different frequencies / azimuth modes, not at all for
    # the lowest frequency, and -- for the substrate case -- so that the substrate disappears from the pruned solves.
    def coarse(seed, L, micro, thick, **kw):
        rng = np.random.default_rng(seed)
        dens = rng.uniform(200, 420, L)
        temp = rng.uniform(235, 268, L)
        th = rng.uniform(0.6 * thick, 1.4 * thick, L)
        if micro == "exponential":
            return make_snowpack(th, micro, density=dens, temperature=temp, corr_length=rng.uniform(1.5e-4, 4e-4, L), **kw)
        return make_snowpack(th, micro, density=dens, temperature=temp, radius=rng.uniform(0.8e-4, 1.8e-4, L),
                             stickiness=0.2, **kw)

    if wanted("iba_L8_n12_prune"):
        spx = coarse(81, 8, "exponential", 0.3)
        save("iba_L8_n12_prune", run_new("iba", passive([10.65e9, 36.5e9, 89e9], [30, 55]), spx,
                                         rtsolver_options=dict(n_max_stream=12, prune_deep_snowpack=2.5)))
    if wanted("iba_L6_n16_prune_substrate"):
        spx = coarse(82, 6, "exponential", 0.3, substrate=Flat(temperature=272.0, permittivity_model=8.0 + 1.0j))
        save("iba_L6_n16_prune_substrate", run_new("iba", passive([18.7e9, 36.5e9, 89e9], [40, 55]), spx,
                                                   rtsolver_options=dict(n_max_stream=16, prune_deep_snowpack=1.3)))
    if wanted("dmrt_L7_n12_prune"):
        spx = coarse(83, 7, "sticky_hard_spheres", 0.4)
        save("dmrt_L7_n12_prune", run_new("dmrt_qca_shortrange", passive([36.5e9, 89e9], [55]), spx,
                                          rtsolver_options=dict(n_max_stream=12, prune_deep_snowpack=3)))
    if wanted("dmrt_L6_n10_prune_over_bad_layer"):
        # layers 5 and 6 have spheres far too large for DMRT at 89 GHz (albedo >= 1: the diagonalisation fails), but the
        # pack is cut after layer 4 there, so the reference never diagonalises them and succeeds
        spx = make_snowpack([0.3, 0.4, 0.35, 0.5, 0.4, 0.6], "sticky_hard_spheres", density=[250, 300, 280, 330, 350, 380],
                            temperature=[250, 252, 255, 258, 260, 262],
                            radius=[1.5e-4, 1.7e-4, 1.6e-4, 1.8e-4, 6e-4, 6e-4], stickiness=0.2)
        save("dmrt_L6_n10_prune_over_bad_layer", run_new("dmrt_qca_shortrange", passive([18.7e9, 89e9], [55]), spx,
                                                         rtsolver_options=dict(n_max_stream=10, prune_deep_snowpack=3)))
    if wanted("iba_active_L6_n10_prune"):
        spx = coarse(84, 6, "exponential", 0.35)
        save("iba_active_L6_n10_prune", run_new("iba", active(17.2e9, [30, 45]), spx,
                                                 rtsolver_options=dict(n_max_stream=10, m_max=2, prune_deep_snowpack=0.45)))

    # Conditioning of the backscatter over a reflecting substrate (found by tools/stress_vs_oracle.py, seed 14): sigma0
    # is what is left (-52 dB) after subtracting a coherent reflection ~1e6 times larger, and the reference's own
    # diagonalisation methods agree to ~1e-8 only.  Stored: the result of each method.
    if wanted("iba_shs_active_substrate_conditioning"):
        spx = make_snowpack([0.2130814222075326, 0.5], "sticky_hard_spheres", density=[346.6822783959207, 197.43486773768632],
                            temperature=[248.17656234012384, 231.0307860458429],
                            radius=[5.676126417172697e-05, 8.92312279601347e-05], stickiness=0.2,
                            substrate=Flat(temperature=264.7614846918132,
                                           permittivity_model=6.012944835080793 + 0.18867742645808616j))
        se = active(5.405e9, [7.785449934645355, 35.627555790446976])
        out = run_new("iba", se, spx, rtsolver_options=dict(n_max_stream=16, m_max=2))
        for meth in ("eig", "half_rank_eig"):
            alt = run_new("iba", se, spx, rtsolver_options=dict(n_max_stream=16, m_max=2, diagonalization_method=meth))
            out["result_" + meth] = alt["result"]
        save("iba_shs_active_substrate_conditioning", out)

    # Heterogeneous snowpacks (smrt/core/model.py:529-582): a list of emmodels, one per layer, over layers that mix the
    # exponential and the sticky-hard-spheres microstructure models -- passive and active
    def mixed_pack(**kw):
        return make_snowpack([0.2, 0.3, 0.25, 10.0], ["exponential", "sticky_hard_spheres", "exponential", "sticky_hard_spheres"],
                             density=[250, 300, 350, 400], temperature=[255, 258, 261, 264],
                             corr_length=[1e-4, None, 2e-4, None], radius=[None, 1.2e-4, None, 1.0e-4],
                             stickiness=[None, 0.2, None, 0.3], **kw)

    if wanted("mixed_L4_n16_passive"):
        save("mixed_L4_n16_passive", run_new(["iba", "dmrt_qca_shortrange", "iba", "iba"], passive([18.7e9, 36.5e9], [40, 55]),
                                              mixed_pack(), rtsolver_options=dict(n_max_stream=16)))
    if wanted("mixed_L4_n12_active"):
        save("mixed_L4_n12_active", run_new(["iba", "dmrt_qca_shortrange", "nonscattering", "iba"], active(13.4e9, [30, 45]),
                                             mixed_pack(), rtsolver_options=dict(n_max_stream=12, m_max=2)))

    # Full-size shapes of BASELINE.json's configs[2] and configs[3] (round 2): a second DMRT snowpack at ALL seven AMSR2
    # frequencies with 64 streams, and the true active shape -- IBA, sentinel1(), 30 thin layers, 128 streams, m_max 2
    # (N = 384 for the azimuth modes 1, 2).
    if wanted("cfg3_dmrt_L50_n64_amsr2_sp1"):
        rng = np.random.default_rng(33)
        spx = random_snowpack(rng, 50, "sticky_hard_spheres")
        save("cfg3_dmrt_L50_n64_amsr2_sp1", run_new("dmrt_qca_shortrange", sensor_list.amsr2(), spx,
                                                     rtsolver_options=dict(n_max_stream=64)))
    for i in range(2):
        name = "cfg4_iba_active_L30_n128_sp%d" % i
        if wanted(name):
            rng = np.random.default_rng(40 + i)
            spx = random_snowpack(rng, 30, "exponential", 0.02, 0.10, 1000.0)
            save(name, run_new("iba", sensor_list.sentinel1(), spx, rtsolver_options=dict(n_max_stream=128, m_max=2)))

    # (iv-c) emmodels WITHOUT a device implementation (evaluated on the host and handed over as numbers): the
    # reference's rayleigh (independent spheres) and prescribed_kskaeps
    if wanted("rayleigh_L3_n16_passive") or wanted("rayleigh_L3_n12_active"):
        spr = make_snowpack([0.2, 0.3, 100.0], "independent_sphere", density=[120.0, 180.0, 250.0],
                            temperature=[255.0, 260.0, 266.0], radius=[2.0e-4, 3.0e-4, 2.5e-4])
        if wanted("rayleigh_L3_n16_passive"):
            save("rayleigh_L3_n16_passive", run_new("rayleigh", sensor_list.passive([18.7e9, 36.5e9], [40.0, 55.0]), spr,
                                                    rtsolver_options=dict(n_max_stream=16)))
        if wanted("rayleigh_L3_n12_active"):
            save("rayleigh_L3_n12_active", run_new("rayleigh", sensor_list.active(13.4e9, [30.0, 45.0]), spr,
                                                   rtsolver_options=dict(n_max_stream=12, m_max=2)))
    if wanted("prescribed_L3_n16_passive"):
        spp = make_snowpack([0.15, 0.4, 50.0], "homogeneous", density=[200.0, 300.0, 350.0], temperature=[250.0, 258.0, 265.0])
        for lay, ks, ka, eps in zip(spp.layers, (0.4, 1.1, 0.7), (0.05, 0.12, 0.2), (1.35 + 2e-4j, 1.6 + 5e-4j, 1.75 + 8e-4j)):
            lay.ks, lay.ka, lay.effective_permittivity = ks, ka, eps
        save("prescribed_L3_n16_passive", run_new("prescribed_kskaeps", sensor_list.passive(36.5e9, [30.0, 55.0]), spp,
                                                  rtsolver_options=dict(n_max_stream=16)))

    # (iv-d) DORT option process_coherent_layers: thin ice lenses / crusts become coherent (Fabry-Perot) interfaces.
    # One frequency per run (the reference's test `if coherent_layers[-1]` breaks on the frequency arrays of a
    # multi-frequency Model.run; single-frequency sensors are what works).  Which layers are coherent depends on the
    # frequency: at 10.65 GHz both the 2 mm crust on top and the 3 mm lens, at 36.5 GHz only ... none of the 3 mm one.
    for name, sens, opts in (
        ("coherent_L5_n16_passive", [sensor_list.passive(f, [40.0, 55.0]) for f in (10.65e9, 18.7e9, 36.5e9)],
         dict(n_max_stream=16)),
        ("coherent_L5_n12_active", [sensor_list.active(f, [30.0, 45.0]) for f in (5.4e9, 13.4e9)],
         dict(n_max_stream=12, m_max=2)),
    ):
        if wanted(name):
            spc = make_snowpack([0.002, 0.25, 0.003, 0.4, 100.0], "exponential", density=[600.0, 250.0, 900.0, 330.0, 400.0],
                                temperature=[262.0, 260.0, 261.0, 264.0, 268.0],
                                corr_length=[2e-5, 1.2e-4, 1e-5, 2.0e-4, 2.5e-4])
            parts = [run_new("iba", se, spc, rtsolver_options=dict(opts, process_coherent_layers=True)) for se in sens]
            out = dict(parts[0])
            out["frequency"] = np.concatenate([p_["frequency"] for p_ in parts])
            out["result"] = np.concatenate([p_["result"] for p_ in parts])
            for i, p_ in enumerate(parts):      # per-frequency diagnostics: the kept layers differ
                for k in ("stream_angles", "effective_permittivity", "ks", "ke", "ka"):
                    out["f%d_%s" % (i, k)] = p_["f0_" + k]
            # and without the option, to show the fixtures are sensitive to it
            out["result_incoherent"] = np.concatenate([run_new("iba", se, spc, rtsolver_options=opts)["result"] for se in sens])
            save(name, out)

    # (iv-d2) process_coherent_layers together with an emmodel that has no device implementation (the reference's rayleigh on
    # independent spheres): the phase matrices of the kept layers live on the streams of the REDUCED snowpack.  The 3 mm
    # lens is coherent at both frequencies, the 6 mm layer only at 10.65 GHz.
    for name, sens, opts in (
        ("rayleigh_coherent_L5_n12_passive", [sensor_list.passive(f, [40.0, 55.0]) for f in (10.65e9, 18.7e9)], dict(n_max_stream=12)),
        ("rayleigh_coherent_L5_n10_active", [sensor_list.active(f, [30.0, 45.0]) for f in (13.4e9,)], dict(n_max_stream=10, m_max=2)),
    ):
        if wanted(name):
            spc = make_snowpack([0.2, 0.003, 0.3, 0.006, 100.0], "independent_sphere", density=[150.0, 800.0, 220.0, 500.0, 300.0],
                                temperature=[255.0, 258.0, 260.0, 262.0, 266.0], radius=[2.0e-4, 1.0e-4, 3.0e-4, 1.5e-4, 2.5e-4])
            parts = [run_new("rayleigh", se, spc, rtsolver_options=dict(opts, process_coherent_layers=True)) for se in sens]
            out = dict(parts[0])
            out["frequency"] = np.concatenate([p_["frequency"] for p_ in parts])
            out["result"] = np.concatenate([p_["result"] for p_ in parts])
            for i, p_ in enumerate(parts):
                for k in ("stream_angles", "effective_permittivity", "ks", "ke", "ka"):
                    out["f%d_%s" % (i, k)] = p_["f0_" + k]
            out["result_incoherent"] = np.concatenate([run_new("rayleigh", se, spc, rtsolver_options=opts)["result"] for se in sens])
            save(name, out)

    # (iv-e) rough substrates in active mode (backscatter of snow over rough soil): the DENSE reflection matrix of the bottom
    # boundary, per azimuth mode, as the reference builds it (rtsolver_utils.py:567-597,690-707: specular diagonal +
    # 2 pi | pi x the weighted diffuse modes) is stored as an INPUT of the fixture -- smrt_amd takes it from the caller
    # (SMRT_SUBSTRATE_HOST) and does not restate the interface physics.  (In passive mode the reference raises inside
    # dort.py:437 for these substrates.)
    from smrt import make_soil
    for name, soil_kw, n_str in (
        ("rough_go_substrate_L2_n12_active", dict(substrate_model="geometrical_optics", mean_square_slope=0.05), 12),
        ("rough_iem_substrate_L3_n10_active", dict(substrate_model="iem_fung92", roughness_rms=0.004, corr_length=0.05), 10),
    ):
        if not wanted(name):
            continue
        soil = make_soil(permittivity_model=complex(8.0, 1.0), temperature=268.0, **soil_kw)
        L = 2 if "L2" in name else 3
        spq = make_snowpack([0.3, 0.25, 0.8][:L], "exponential", density=[250.0, 300.0, 350.0][:L],
                            temperature=[258.0, 261.0, 264.0][:L], corr_length=[1e-4, 1.5e-4, 2e-4][:L], substrate=soil)
        sens = sensor_list.active(13.4e9, [30.0, 40.0])
        opts = dict(n_max_stream=n_str, m_max=2)
        SKIP_OLD[0] = False
        mq = make_model("iba", "dort", rtsolver_options=opts)
        sims, _ = mq.prepare_simulations(sens, spq, None, "snowpack")
        sef, spk = list(sims)[0]
        res = mq.run_single_simulation((sef, spk), None, None)
        ems = mq.prepare_emmodels(sef, spk)
        solver = DORT(**opts)
        solver.init_solve(spk, ems, sef, None)
        solver.prepare_streams()
        itf = compute_interface_properties(sef.frequency, spk.interfaces, spk.substrate, solver.effective_permittivity,
                                           solver.streams, opts["m_max"], 3)
        out = dict(snowpack_arrays(spk))
        out.update(emmodel="iba", mode="A", frequency=np.array([float(sef.frequency)]), result=np.asarray(res.data.values)[None],
                   theta_deg=np.asarray(sens.theta_deg, float), theta_inc_deg=np.asarray(sens.theta_inc_deg, float),
                   opt_n_max_stream=n_str, opt_m_max=2, substrate_kind="host", substrate_temperature=268.0,
                   streams_n=np.asarray(solver.streams.n, int))
        nb = int(solver.streams.n[L - 1])
        for mode in range(opts["m_max"] + 1):
            P = 2 if mode == 0 else 3
            def dense(x):   # ndarray, the reference's diagonal wrapper (smrt_diag) or the scalar 0
                if hasattr(x, "diagonal") and type(x).__name__ == "smrt_diag":
                    return np.diag(np.asarray(x.diagonal(), float))
                x = np.asarray(x, float)
                return np.zeros((nb * P, nb * P)) if x.ndim == 0 else (np.diag(x) if x.ndim == 1 else x)
            R = dense(itf.reflection_bottom(L - 1, mode, False))
            Rc = np.diag(dense(itf.reflection_bottom(L - 1, mode, True))).copy()
            assert R.shape == (nb * P, nb * P) and Rc.shape == (nb * P,), (name, mode, R.shape, Rc.shape, nb, P)
            out["sub_R_m%d" % mode] = R
            out["sub_Rcoh_m%d" % mode] = Rc
        for k in ("stream_angles", "effective_permittivity", "ks", "ke", "ka"):
            out["f0_" + k] = np.asarray(res.other_data[k].values)
        # what the substrate object itself returns on the streams of the last layer (the protocol a host-side substrate
        # speaks, rtsolver_utils.py:567-597): specular diagonal [3, n] and, if any, the raw diffuse modes [3, 3, m, n, n]
        mu_b = np.asarray(solver.streams.mu[L - 1], float)
        eps_b = solver.effective_permittivity[L - 1]
        out["sub_mu"] = mu_b
        out["sub_weight"] = np.asarray(solver.streams.weight[L - 1], float)
        spec = spk.substrate.specular_reflection_matrix(sef.frequency, eps_b, mu_b, 3)
        spec = np.asarray(getattr(spec, "values", spec), float)
        out["sub_spec_raw"] = np.zeros((3, nb)) if spec.ndim == 0 else spec.reshape(3, nb)
        raw = spk.substrate.ft_even_diffuse_reflection_matrix(sef.frequency, eps_b, mu_b, mu_b, opts["m_max"], 3)
        out["sub_diff_mtype"] = str(getattr(raw, "mtype", "dense5"))
        out["sub_diff_raw"] = np.asarray(getattr(raw, "values", raw), float)
        SKIP_OLD[0] = bool(ONLY)
        save(name, out)

    # (iv-f) the rough substrates the reference does run in PASSIVE mode (iem_fung92: specular + diagonal diffuse part;
    # geometrical_optics_backscatter): dense bottom reflection of mode 0 and the emissivity diagonal as fixture inputs
    for name, soil_kw, n_str in (
        ("rough_iem_substrate_L3_n10_passive", dict(substrate_model="iem_fung92", roughness_rms=0.004, corr_length=0.05), 10),
        ("rough_gob_substrate_L2_n12_passive", dict(substrate_model="geometrical_optics_backscatter", mean_square_slope=0.05), 12),
    ):
        if not wanted(name):
            continue
        soil = make_soil(permittivity_model=complex(8.0, 1.0), temperature=268.0, **soil_kw)
        L = 2 if "L2" in name else 3
        spq = make_snowpack([0.3, 0.25, 0.8][:L], "exponential", density=[250.0, 300.0, 350.0][:L],
                            temperature=[258.0, 261.0, 264.0][:L], corr_length=[1e-4, 1.5e-4, 2e-4][:L], substrate=soil)
        sens = sensor_list.passive(18.7e9, [40.0, 55.0])
        opts = dict(n_max_stream=n_str)
        SKIP_OLD[0] = False
        mq = make_model("iba", "dort", rtsolver_options=opts)
        sims, _ = mq.prepare_simulations(sens, spq, None, "snowpack")
        sef, spk = list(sims)[0]
        res = mq.run_single_simulation((sef, spk), None, None)
        ems = mq.prepare_emmodels(sef, spk)
        solver = DORT(**opts)
        solver.init_solve(spk, ems, sef, None)
        solver.prepare_streams()
        itf = compute_interface_properties(sef.frequency, spk.interfaces, spk.substrate, solver.effective_permittivity,
                                           solver.streams, 0, 2)
        nb = int(solver.streams.n[L - 1])

        def dense2(x):
            if type(x).__name__ == "smrt_diag":
                return np.diag(np.asarray(x.diagonal(), float))
            x = np.asarray(x, float)
            return np.zeros((nb * 2, nb * 2)) if x.ndim == 0 else (np.diag(x) if x.ndim == 1 else x)
        out = dict(snowpack_arrays(spk))
        out.update(emmodel="iba", mode="P", frequency=np.array([float(sef.frequency)]), result=np.asarray(res.data.values)[None],
                   theta_deg=np.asarray(sens.theta_deg, float), opt_n_max_stream=n_str, substrate_kind="host",
                   substrate_temperature=268.0, streams_n=np.asarray(solver.streams.n, int),
                   sub_R_m0=dense2(itf.reflection_bottom(L - 1, 0, False)),
                   sub_emis=np.diag(dense2(itf.transmission_bottom(L - 1, 0, False))).copy())
        mu_b = np.asarray(solver.streams.mu[L - 1], float)
        eps_b = solver.effective_permittivity[L - 1]
        out["sub_mu"], out["sub_weight"] = mu_b, np.asarray(solver.streams.weight[L - 1], float)
        for key, val in (("sub_spec_raw", spk.substrate.specular_reflection_matrix(sef.frequency, eps_b, mu_b, 2)),
                         ("sub_emis_raw", spk.substrate.emissivity_matrix(sef.frequency, eps_b, mu_b, 2)),
                         ("sub_diff_raw", spk.substrate.ft_even_diffuse_reflection_matrix(sef.frequency, eps_b, mu_b, mu_b, 0, 2))):
            val = np.asarray(getattr(val, "values", val), float)
            out[key] = np.zeros((2, nb)) if val.ndim == 0 and key != "sub_diff_raw" else val
        for k in ("stream_angles", "effective_permittivity", "ks", "ke", "ka"):
            out["f0_" + k] = np.asarray(res.other_data[k].values)
        SKIP_OLD[0] = bool(ONLY)
        save(name, out)

    # (iv-g) ROUGH INTERFACES at the surface / between layers (rtsolver_utils.py:473-642, interface/iem_fung92.py,
    # interface/geometrical_optics.py): like the rough substrates, the interface physics stays with the caller -- the
    # fixtures hold, per rough interface i (on top of layer i, 0 = the surface) and azimuth mode, the four dense matrices
    # the reference combines (Rtop / Ttop of layer i looking up, Rbot / Tbot of the medium above looking down; and their
    # specular-only versions of the coherent pass) as INPUTS, plus what the interface object itself returned on the stream
    # grids (the protocol a host-side interface speaks).  Passive: iem_fung92 (geometrical_optics does not conserve energy
    # in passive mode -- the reference itself warns); active: both.
    from smrt import make_interface
    from smrt.interface.flat import Flat as RefFlat
    for name, iname, ikw, where, mode_, n_str in (
        ("rough_iem_surface_L3_n10_passive", "iem_fung92", dict(roughness_rms=0.002, corr_length=0.05), 0, "P", 10),
        ("rough_iem_inner_L3_n10_passive", "iem_fung92", dict(roughness_rms=0.002, corr_length=0.05), 2, "P", 10),
        ("rough_go_surface_L3_n10_active", "geometrical_optics", dict(mean_square_slope=0.03), 0, "A", 10),
        ("rough_iem_inner_L3_n10_active", "iem_fung92", dict(roughness_rms=0.002, corr_length=0.05), 2, "A", 10),
    ):
        if not wanted(name):
            continue
        rough = make_interface(iname, **ikw)
        ilist = [RefFlat, RefFlat, RefFlat]
        ilist[where] = rough
        spq = make_snowpack([0.3, 0.25, 100.0], "exponential", density=[250.0, 300.0, 350.0], temperature=[258.0, 261.0, 264.0],
                            corr_length=[1e-4, 1.5e-4, 2e-4], interface=ilist)
        act = mode_ == "A"
        sens = sensor_list.active(13.4e9, [30.0, 40.0]) if act else sensor_list.passive(18.7e9, [40.0, 55.0])
        opts = dict(n_max_stream=n_str, m_max=2) if act else dict(n_max_stream=n_str)
        SKIP_OLD[0] = False
        mq = make_model("iba", "dort", rtsolver_options=opts)
        sims, _ = mq.prepare_simulations(sens, spq, None, "snowpack")
        sef, spk = list(sims)[0]
        res = mq.run_single_simulation((sef, spk), None, None)
        ems = mq.prepare_emmodels(sef, spk)
        solver = DORT(**opts)
        solver.init_solve(spk, ems, sef, None)
        solver.prepare_streams()
        m_max, npol = (2, 3) if act else (0, 2)
        itf = compute_interface_properties(sef.frequency, spk.interfaces, spk.substrate, solver.effective_permittivity,
                                           solver.streams, m_max, npol)
        st = solver.streams
        n_low = int(st.n[where])
        n_up = int(st.n[where - 1]) if where > 0 else int(len(st.outmu))
        out = dict(snowpack_arrays(spk))
        out.update(emmodel="iba", mode=mode_, frequency=np.array([float(sef.frequency)]), result=np.asarray(res.data.values)[None],
                   theta_deg=np.asarray(sens.theta_deg, float), opt_n_max_stream=n_str, streams_n=np.asarray(st.n, int),
                   rough_interface=np.array([where]))
        if act:
            out.update(theta_inc_deg=np.asarray(sens.theta_inc_deg, float), opt_m_max=2)

        def densify(x, rows, cols):
            if type(x).__name__ == "smrt_diag":
                x = np.asarray(x.diagonal(), float)
            x = np.asarray(getattr(x, "values", x), float)
            if x.ndim == 0:
                return np.zeros((rows, cols))
            return np.diag(x) if x.ndim == 1 else x
        up_idx = where - 1 if where > 0 else -1
        for m in range(m_max + 1):
            P = 2 if m == 0 else 3
            for coh in (False, True):
                tag = "_coh" if coh else ""
                out["itf%d_Rtop%s_m%d" % (where, tag, m)] = densify(itf.reflection_top(where, m, coh), n_low * P, n_low * P)
                out["itf%d_Ttop%s_m%d" % (where, tag, m)] = densify(itf.transmission_top(where, m, coh), n_low * P, n_low * P)
                out["itf%d_Rbot%s_m%d" % (where, tag, m)] = densify(itf.reflection_bottom(up_idx, m, coh), n_up * P, n_up * P)
                out["itf%d_Tbot%s_m%d" % (where, tag, m)] = densify(itf.transmission_bottom(up_idx, m, coh), n_up * P, n_up * P)
        # the protocol outputs of the interface object on the grids compute_interface_properties uses (rtsolver_utils.py:
        # 476-642; NB its mu_t of the upward diffuse transmission is streams.mu[layer - 1] only for layer > 1)
        eps = solver.effective_permittivity
        e_l, e_u = eps[where], (eps[where - 1] if where > 0 else 1)
        mu_l = np.asarray(st.mu[where], float)
        mu_u = np.asarray(st.mu[where - 1], float) if where > 0 else np.asarray(st.outmu, float)
        mu_t_up = np.asarray(st.mu[where - 1], float) if where > 1 else np.asarray(st.outmu, float)
        w_l = np.asarray(st.weight[where], float)
        w_u = np.asarray(st.weight[where - 1], float) if where > 0 else np.asarray(st.outweight, float)
        out.update(itf_mu_low=mu_l, itf_mu_up=mu_u, itf_mu_t_up=mu_t_up, itf_w_low=w_l, itf_w_up=w_u,
                   itf_eps_low=np.array([complex(e_l)]), itf_eps_up=np.array([complex(e_u)]))
        raw = dict(
            spec_up=rough.specular_reflection_matrix(sef.frequency, e_l, e_u, mu_l, npol),
            spec_dn=rough.specular_reflection_matrix(sef.frequency, e_u, e_l, mu_u, npol),
            ctr_up=rough.coherent_transmission_matrix(sef.frequency, e_l, e_u, mu_l, npol),
            ctr_dn=rough.coherent_transmission_matrix(sef.frequency, e_u, e_l, mu_u, npol))
        if hasattr(rough, "ft_even_diffuse_reflection_matrix"):
            raw.update(drf_up=rough.ft_even_diffuse_reflection_matrix(sef.frequency, e_l, e_u, mu_l, mu_l, m_max, npol),
                       drf_dn=rough.ft_even_diffuse_reflection_matrix(sef.frequency, e_u, e_l, mu_u, mu_u, m_max, npol))
        if hasattr(rough, "ft_even_diffuse_transmission_matrix"):   # (iem_fung92 has no diffuse transmission)
            raw.update(dtr_up=rough.ft_even_diffuse_transmission_matrix(sef.frequency, e_l, e_u, mu_t_up, mu_l, m_max, npol),
                       dtr_dn=rough.ft_even_diffuse_transmission_matrix(sef.frequency, e_u, e_l, mu_l, mu_u, m_max, npol))
        for k, v in raw.items():
            out["itf_raw_" + k + "_mtype"] = str(getattr(v, "mtype", "none"))
            out["itf_raw_" + k] = np.asarray(getattr(v, "values", v), float)
        for k in ("stream_angles", "effective_permittivity", "ks", "ke", "ka"):
            out["f0_" + k] = np.asarray(res.other_data[k].values)
        SKIP_OLD[0] = bool(ONLY)
        save(name, out)

    # (iv-h) IBA with emmodel_options=dict(dense_snow_correction="auto") (smrt/emmodel/iba.py:85-105): layers above half ice
    # are modelled as air inclusions in ice (core/layer.py:186-201), the others are left alone.  Firn / ice-lens like
    # densities next to ordinary snow, both microstructure models, passive and active.
    if wanted("iba_dense_auto_L5_n12"):
        spx = make_snowpack([0.15, 0.3, 0.2, 0.5, 100.0], "exponential", density=[300, 550, 700, 850, 400],
                            temperature=[255, 258, 260, 262, 265], corr_length=[1e-4, 2.5e-4, 3e-4, 2e-4, 1.5e-4])
        save("iba_dense_auto_L5_n12", run_new("iba", passive([18.7e9, 36.5e9, 89e9], [30, 55]), spx,
                                               rtsolver_options=dict(n_max_stream=12), stages=True, stage_layers=(2,),
                                               emmodel_options=dict(dense_snow_correction="auto"),
                                               emmodel_label="iba_dense_auto"))
    if wanted("iba_dense_auto_shs_active_L3_n8"):
        spx = make_snowpack([0.2, 0.4, 1000.0], "sticky_hard_spheres", density=[600, 350, 800], temperature=[258, 262, 266],
                            radius=[2e-4, 1.5e-4, 2.5e-4], stickiness=0.3)
        save("iba_dense_auto_shs_active_L3_n8", run_new("iba", active(13.4e9, [30, 45]), spx,
                                                         rtsolver_options=dict(n_max_stream=8, m_max=2),
                                                         emmodel_options=dict(dense_snow_correction="auto"),
                                                         emmodel_label="iba_dense_auto"))

    # (iv-i) WET SNOW (smrt/inputs/make_medium.py:316-434, smrt/permittivity/wetice.py:12-45, water.py:14-43): layers at the
    # melting point holding liquid water -- given as volumetric_liquid_water and as liquid_water --, the grains' permittivity
    # by Maxwell Garnett in a water host; IBA passive / active (one very wet layer above half "ice + water": inverted by
    # dense_snow_correction="auto") and DMRT
    if wanted("iba_wet_L4_n12_passive"):
        spx = make_snowpack([0.1, 0.25, 0.4, 100.0], "exponential", density=[250, 320, 380, 420],
                            temperature=[273.15, 273.15, 268.0, 265.0], corr_length=[1.2e-4, 2e-4, 1.8e-4, 1.5e-4],
                            volumetric_liquid_water=[0.03, 0.005, 0.0, 0.0])
        save("iba_wet_L4_n12_passive", run_new("iba", passive([6.925e9, 18.7e9, 36.5e9], [40, 55]), spx,
                                                rtsolver_options=dict(n_max_stream=12), stages=True, stage_layers=(0,)))
    if wanted("iba_wet_L3_n10_active"):
        spx = make_snowpack([0.05, 0.3, 1000.0], "sticky_hard_spheres", density=[300, 700, 400],
                            temperature=[273.15, 273.15, 266.0], radius=[4e-4, 5e-4, 3e-4], stickiness=0.3,
                            liquid_water=[0.003, 0.01, 0.0])
        save("iba_wet_L3_n10_active", run_new("iba", active(13.4e9, [30, 40]), spx,
                                               rtsolver_options=dict(n_max_stream=10, m_max=2),
                                               emmodel_options=dict(dense_snow_correction="auto"),
                                               emmodel_label="iba_dense_auto"))
    if wanted("dmrt_wet_L3_n12_passive"):
        spx = make_snowpack([0.2, 0.5, 100.0], "sticky_hard_spheres", density=[280, 350, 400],
                            temperature=[273.15, 270.0, 266.0], radius=[1.5e-4, 2e-4, 1.2e-4], stickiness=0.2,
                            volumetric_liquid_water=[0.02, 0.0, 0.0])
        save("dmrt_wet_L3_n12_passive", run_new("dmrt_qca_shortrange", passive([10.65e9, 36.5e9], [55]), spx,
                                                 rtsolver_options=dict(n_max_stream=12)))

    # (iv-j) IBA over the other closed-form microstructure models (smrt/microstructure_model/teubner_strey.py:45-55,
    # independent_sphere.py:54-72) mixed with the two usual ones, one model per layer, passive and active
    micro4 = ["teubner_strey", "independent_sphere", "exponential", "sticky_hard_spheres"]
    if wanted("iba_micro4_L4_n12_passive") or wanted("iba_micro4_L4_n10_active"):
        def micro_pack(last):
            return make_snowpack([0.2, 0.3, 0.4, last], micro4, density=[280, 320, 360, 400], temperature=[258, 261, 264, 266],
                                 corr_length=[1.5e-4, None, 2e-4, None], repeat_distance=[1.2e-3, None, None, None],
                                 radius=[None, 2.5e-4, None, 2e-4], stickiness=[None, None, None, 0.3])
        if wanted("iba_micro4_L4_n12_passive"):
            save("iba_micro4_L4_n12_passive", run_new("iba", passive([18.7e9, 36.5e9, 89e9], [40, 55]), micro_pack(100.0),
                                                       rtsolver_options=dict(n_max_stream=12)))
        if wanted("iba_micro4_L4_n10_active"):
            save("iba_micro4_L4_n10_active", run_new("iba", active(13.4e9, [30, 40]), micro_pack(1000.0),
                                                      rtsolver_options=dict(n_max_stream=10, m_max=2)))

    # (iv-k) IBA over the models on the unified parameters (smrt/microstructure_model/unified_scaled_exponential.py,
    # unified_teubner_strey.py on both sides of polydispersity 1, unified_sticky_hard_spheres.py), one per layer
    unified4 = ["unified_scaled_exponential", "unified_teubner_strey", "unified_teubner_strey", "unified_sticky_hard_spheres"]
    if wanted("iba_unified4_L4_n12_passive") or wanted("iba_unified4_L4_n10_active"):
        def unified_pack(last):
            return make_snowpack([0.2, 0.3, 0.4, last], unified4, density=[280, 320, 360, 400], temperature=[258, 261, 264, 266],
                                 porod_length=[1.2e-4, 1.5e-4, 1.1e-4, 1.6e-4], polydispersity=[0.8, 0.7, 1.6, 1.2])
        if wanted("iba_unified4_L4_n12_passive"):
            save("iba_unified4_L4_n12_passive", run_new("iba", passive([18.7e9, 36.5e9, 89e9], [40, 55]), unified_pack(100.0),
                                                         rtsolver_options=dict(n_max_stream=12)))
        if wanted("iba_unified4_L4_n10_active"):
            save("iba_unified4_L4_n10_active", run_new("iba", active(13.4e9, [30, 40]), unified_pack(1000.0),
                                                        rtsolver_options=dict(n_max_stream=10, m_max=2)))

    # (iv-l) the other members of IBA's family (smrt/emmodel/iba_original.py, iba_maxwell_garnett.py): IBA's phase matrix with
    # other scalars -- on the device through SMRT_EM_IBA_HOST (the scalars from the emmodel object, the phase matrices
    # assembled by the kernels); one emmodel for the snowpack, and one per layer mixed with plain IBA
    if wanted("iba_original_L3_n12_passive") or wanted("iba_mg_L3_n10_active") or wanted("iba_family_L3_n12_passive"):
        def family_pack(last, micro="exponential"):
            return make_snowpack([0.25, 0.35, last], micro, density=[220, 310, 390], temperature=[257, 262, 266],
                                 corr_length=[1.2e-4, 2.0e-4, 2.6e-4]) if micro == "exponential" else \
                make_snowpack([0.25, 0.35, last], micro, density=[220, 310, 390], temperature=[257, 262, 266],
                              radius=[1.5e-4, 2.0e-4, 2.4e-4], stickiness=[0.2, 0.3, 0.5])
        if wanted("iba_original_L3_n12_passive"):
            save("iba_original_L3_n12_passive", run_new("iba_original", passive([10.65e9, 36.5e9, 89e9], [40, 55]), family_pack(100.0),
                                                         rtsolver_options=dict(n_max_stream=12)))
        if wanted("iba_mg_L3_n10_active"):
            save("iba_mg_L3_n10_active", run_new("iba_maxwell_garnett", active(13.4e9, [30, 40]), family_pack(1000.0, "sticky_hard_spheres"),
                                                  rtsolver_options=dict(n_max_stream=10, m_max=2)))
        if wanted("iba_family_L3_n12_passive"):
            save("iba_family_L3_n12_passive", run_new(["iba_maxwell_garnett", "iba", "iba_original"], passive([18.7e9, 36.5e9], [55]),
                                                       family_pack(100.0), rtsolver_options=dict(n_max_stream=12)))

    # (iv-m) the symmetrised strong-contrast expansion (smrt/emmodel/symsce_torquato21.py): IBA's phase function at a complex
    # wavenumber; passive, on the exponential model and on its unified parametrisation
    if wanted("symsce_L3_n12_passive") or wanted("symsce_unified_L3_n12_passive"):
        if wanted("symsce_L3_n12_passive"):
            spx = make_snowpack([0.25, 0.35, 100.0], "exponential", density=[220, 310, 390], temperature=[257, 262, 266],
                                corr_length=[1.2e-4, 2.0e-4, 2.6e-4])
            save("symsce_L3_n12_passive", run_new("symsce_torquato21", passive([10.65e9, 36.5e9, 89e9], [40, 55]), spx,
                                                   rtsolver_options=dict(n_max_stream=12)))
        if wanted("symsce_unified_L3_n12_passive"):
            spx = make_snowpack([0.25, 0.35, 100.0], "unified_scaled_exponential", density=[220, 310, 390], temperature=[257, 262, 266],
                                porod_length=[1.0e-4, 1.6e-4, 2.0e-4], polydispersity=[1.2, 1.1, 1.3])
            save("symsce_unified_L3_n12_passive", run_new("symsce_torquato21", passive([18.7e9, 36.5e9], [55]), spx,
                                                           rtsolver_options=dict(n_max_stream=12)))

    if wanted("symsce_ts_L3_n12_passive"):   # ... and on Teubner-Strey's expression, unified parameters on both sides of polydispersity 1
        spx = make_snowpack([0.25, 0.35, 100.0], "unified_teubner_strey", density=[220, 310, 390], temperature=[257, 262, 266],
                            porod_length=[1.0e-4, 1.6e-4, 2.0e-4], polydispersity=[0.8, 1.4, 1.1])
        save("symsce_ts_L3_n12_passive", run_new("symsce_torquato21", passive([18.7e9, 36.5e9, 89e9], [55]), spx,
                                                  rtsolver_options=dict(n_max_stream=12)))

    if wanted("symsce_spheres_L3_n12_passive"):   # ... and on the sphere models: sines of a complex argument
        spx = make_snowpack([0.25, 0.35, 100.0], ["sticky_hard_spheres", "unified_sticky_hard_spheres", "independent_sphere"],
                            density=[220, 310, 250], temperature=[257, 262, 266], radius=[1.5e-4, None, 2.2e-4],
                            stickiness=[0.2, None, None], porod_length=[None, 1.6e-4, None], polydispersity=[None, 1.3, None])
        save("symsce_spheres_L3_n12_passive", run_new("symsce_torquato21", passive([18.7e9, 36.5e9, 89e9], [55]), spx,
                                                       rtsolver_options=dict(n_max_stream=12)))

    # (v) IBA ks table, smrt/emmodel/test_iba.py:111-127 (shs snowpack of setup_func_pc) and the stream-angle
    # known answer smrt/rtsolver/test_rtsolver.py:64-73
    from smrt.emmodel.iba import IBA

    ks_tab = []
    pcs = [0.3e-3, 0.25e-3, 0.2e-3, 0.15e-3, 0.1e-3, 0.05e-3]
    for pc in pcs:
        spk = make_snowpack([1], "exponential", density=[300], temperature=[265], corr_length=[pc])
        e = IBA(sensor_list.amsre("37V"), spk.layers[0])
        ks_tab.append([pc, e._ks, e.ka, e.effective_permittivity().real, e.effective_permittivity().imag,
                       e.iba_coeff])
    save("iba_ks_table", dict(table=np.array(ks_tab), frequency=36.5e9, density=300.0, temperature=265.0,
                              memls_reference=np.array([4.14237510549, 2.58473097058, 1.41504051e00,
                                                        0.630947615752, 0.194948835313, 0.0250132475909])))

    # (vi) simulation order and result layout of Model.run for 3 snowpacks x AMSR-E (frequency-major)
    m = make_model("iba", "dort")
    sims, dims = m.prepare_simulations(sensor_list.amsre(), sps, None, "snowpack")
    order = [(float(se.frequency), sps.index(s)) for se, s in sims]
    save("model_run_order", dict(order=np.array(order), dims=np.array([str(dm[0]) for dm in dims])))


if __name__ == "__main__":
    main()
