#!/usr/bin/env python
"""Randomised parity sweep on the GPU box (not part of the test-suite: ~2 minutes of oracle time): random snowpacks,
emmodels, stream counts that exercise every kernel path (LDS pipeline, global-workspace pipeline, scalar kernel),
passive / active, with and without substrate and atmosphere, ragged layer counts -- every pair against the CPU oracle.

    python tools/stress_vs_oracle.py [seed] [prune|coherent|wetmicro]
        "prune": also draw a prune_deep_snowpack threshold per case
        "coherent": process_coherent_layers on, with millimetre-thin layers drawn into the snowpacks (some of them last or
        in a row: the refusals of the reference must come back as status 6 for exactly those pairs)
        "wetmicro": in the IBA cases every layer draws its own microstructure model among the four the device has
        (exponential, sticky_hard_spheres, teubner_strey, independent_sphere) and the top layers may be wet (melting
        point, liquid water 0.1 ... 5 %)
        "family": in the IBA cases every layer draws its emmodel among iba, iba_original and iba_maxwell_garnett; the
        last two reach the device as SMRT_EM_IBA_HOST layers -- effective permittivity, ks, ka and the phase coefficient
        from the oracle's layer objects (what the emmodel objects hand over in the product), the phase matrices
        assembled by the kernels
        "sce": the passive IBA cases on the exponential model run the symmetrised strong-contrast expansion instead
        (symsce_torquato21: SMRT_EM_IBA_HOST layers on SMRT_MS_EXPONENTIAL_COMPLEX_K, scalars and phase norm from the
        oracle's restatement) -- the complex-wavenumber branch of the phase assembly on every pipeline
"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import dort_oracle as O  # checker only
from smrt_amd._native import DortContext, PackedBatch

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
with_prune = len(sys.argv) > 2 and sys.argv[2] == "prune"
with_coherent = len(sys.argv) > 2 and sys.argv[2] == "coherent"
with_wetmicro = len(sys.argv) > 2 and sys.argv[2] == "wetmicro"
with_family = len(sys.argv) > 2 and sys.argv[2] == "family"
with_sce = len(sys.argv) > 2 and sys.argv[2] == "sce"
FAMILY = ["iba", "iba_original", "iba_maxwell_garnett", "symsce_torquato21"]
n_family = 0
MS_NAMES = ["exponential", "sticky_hard_spheres", "independent_sphere", "teubner_strey"]   # = MS codes 0 .. 3
n_wet = n_conditioned = 0
n_coherent = n_refused = 0
rng_prune = np.random.default_rng(1000 + (int(sys.argv[1]) if len(sys.argv) > 1 else 1))   # keeps the snowpack stream intact
n_pruned = 0
if os.environ.get("STRESS_EMU"):   # build container: the device source under the CPU emulator (tests/conftest.py), small cases only
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import EmulatedContext
    ctx = EmulatedContext()
else:
    ctx = DortContext(0)
worst_tb, worst_co, worst_cx, n_checked = 0.0, 0.0, 0.0, 0
case_co = case_cx = case_ratio = 0.0
t0 = time.time()
cases = []
for n in (6, 11, 16, 21, 32):                    # passive: N = 12 .. 64 (LDS pipeline)
    cases.append(("P", n, "iba", "exponential"))
cases += [("P", 40, "iba", "exponential"), ("P", 64, "dmrt_qca_shortrange", "sticky_hard_spheres"),   # global-workspace pipeline
          ("P", 70, "iba", "sticky_hard_spheres"),                                                    # scalar kernel (N = 140)
          ("P", 24, "dmrt_qcacp_shortrange", "sticky_hard_spheres"), ("P", 20, "nonscattering", "exponential"),
          ("A", 8, "iba", "exponential"), ("A", 16, "iba", "sticky_hard_spheres"), ("A", 21, "dmrt_qca_shortrange", "sticky_hard_spheres"),
          ("A", 30, "iba", "exponential"), ("A", 42, "iba", "exponential")]
for mode, n, em, ms in cases:
    if os.environ.get("STRESS_EMU") and n > int(os.environ["STRESS_EMU"]):
        continue
    S, Lmax = 4, int(rng.integers(2, 9))
    nl = rng.integers(1, Lmax + 1, S).astype(np.int32); nl[0] = Lmax
    thick = rng.uniform(0.03, 0.4, (S, Lmax)); 
    for s in range(S): thick[s, nl[s] - 1] = rng.choice([0.5, 100.0])
    if with_coherent:   # thin crusts / lenses: a few millimetres (coherent at some of the frequencies only)
        thin = rng_prune.random((S, Lmax)) < 0.3
        thick = np.where(thin, rng_prune.uniform(0.0005, 0.006, (S, Lmax)), thick)
    dens = rng.uniform(150, 450, (S, Lmax)); temp = rng.uniform(230, 270, (S, Lmax))
    if ms == "exponential":
        p1 = rng.uniform(5e-5, 3e-4, (S, Lmax)); p2 = None
    else:
        p1 = rng.uniform(5e-5, 1.5e-4, (S, Lmax)); p2 = np.full((S, Lmax), 0.2)
    freqs = np.sort(rng.choice([6.925e9, 10.65e9, 18.7e9, 36.5e9, 89e9], 2, replace=False)) if mode == "P" else np.array([rng.choice([5.405e9, 13.4e9, 17.2e9])])
    theta = np.sort(rng.uniform(5, 65, 2))
    sub = atm = None
    if rng.random() < 0.5:
        eps = complex(rng.uniform(3, 8), rng.uniform(0.1, 1.0)); Ts = rng.uniform(255, 272, S)
        sub = ("flat", np.full((len(freqs), S), eps.real), np.full((len(freqs), S), eps.imag), Ts)
    if mode == "P" and rng.random() < 0.5:
        atm = (rng.uniform(5, 40, len(freqs)), rng.uniform(2, 15, len(freqs)), rng.uniform(0.8, 1.0, len(freqs)))
    prune = None
    if with_prune and n * (3 if mode == "A" else 2) <= 128:   # the option needs a pipeline
        prune = [0.3, 1.0, 3.0, True][int(rng_prune.integers(0, 4))]
    kinds = lw = msl = None
    if with_wetmicro and em == "iba":
        msl = rng_prune.integers(0, 4, (S, Lmax))                       # microstructure model of every layer
        size = np.where((msl == 0) | (msl == 3), rng_prune.uniform(5e-5, 3e-4, (S, Lmax)), rng_prune.uniform(5e-5, 1.5e-4, (S, Lmax)))
        second = np.where(msl == 1, 0.2, np.where(msl == 3, rng_prune.uniform(5e-4, 3e-3, (S, Lmax)), 0.0))
        p1, p2 = size, np.where(msl == 3, (2 * np.pi * size / np.where(msl == 3, second, 1.0)) ** 2, second)   # Teubner-Strey: micro_p2 = Y
        kinds = 16 * msl                                                  # emmodel code 0 = iba
        lw = np.zeros((S, Lmax))
        for s in range(S):
            for l in range(min(int(rng_prune.integers(0, 3)), nl[s])):   # 0 .. 2 wet layers at the top
                lw[s, l] = 10.0 ** rng_prune.uniform(-3, -1.3); temp[s, l] = 273.15
        n_wet += int((lw > 0).any(axis=1).sum()) * len(freqs)
    host_scalars = eml = None
    if with_family and em == "iba":
        from smrt_amd._native import EM_CODES, MS_CODES
        eml = rng_prune.integers(0, 3, (S, Lmax))                      # emmodel of every layer: index into FAMILY[:3]
        hl = np.zeros((len(freqs), S, Lmax, 4)); hl[..., 2] = 1.0
        hc = np.zeros((len(freqs), S, Lmax))
        for fi, fr in enumerate(freqs):
            for s in range(S):
                k = nl[s]
                spd = dict(thickness=thick[s, :k], density=dens[s, :k], temperature=temp[s, :k], microstructure=ms)
                spd.update(dict(corr_length=p1[s, :k]) if ms == "exponential" else dict(radius=p1[s, :k], stickiness=p2[s, :k]))
                for l, lay in enumerate(O.make_layers([FAMILY[c] for c in eml[s, :k]], float(fr), spd)):
                    hl[fi, s, l] = lay.ks, lay.ka, complex(lay.eps_eff).real, complex(lay.eps_eff).imag
                    hc[fi, s, l] = lay.iba_coeff
        host_scalars = (hl, hc)
        kinds = np.where(eml == 0, EM_CODES["iba"], EM_CODES["iba_host"]) + 16 * MS_CODES[ms]
        n_family += int((eml[:, 0] > 0).sum()) * len(freqs)
    if with_sce and em == "iba" and ms == "exponential" and mode == "P":
        from smrt_amd._native import EM_CODES, MS_CODES
        eml = np.full((S, Lmax), 3)                                    # FAMILY[3]: symsce_torquato21 on every layer
        hl = np.zeros((len(freqs), S, Lmax, 4)); hl[..., 2] = 1.0
        hc = np.zeros((len(freqs), S, Lmax))
        for fi, fr in enumerate(freqs):
            for s in range(S):
                k = nl[s]
                spd = dict(thickness=thick[s, :k], density=dens[s, :k], temperature=temp[s, :k], microstructure=ms, corr_length=p1[s, :k])
                for l, lay in enumerate(O.make_layers("symsce_torquato21", float(fr), spd)):
                    hl[fi, s, l] = lay.ks, lay.ka, complex(lay.eps_eff).real, complex(lay.eps_eff).imag
                    hc[fi, s, l] = lay.iba_coeff
        host_scalars = (hl, hc)
        kinds = np.full((S, Lmax), EM_CODES["iba_host"] + 16 * MS_CODES["exponential_complex_k"])
        n_family += S * len(freqs)
    b = PackedBatch(nl, thick, dens / 916.7, temp, p1, p2, freqs, np.deg2rad(theta), emmodel=em, microstructure=ms, mode=mode,
                    n_max_stream=n, m_max=2, substrate=sub, atmosphere=atm, prune_deep_snowpack=prune,
                    process_coherent_layers=with_coherent, layer_kind=kinds, liquid_water=lw, host_scalars=host_scalars)
    out = ctx.run(b)
    case_co = case_cx = 0.0; case_ratio = 1.0
    for f in range(len(freqs)):
        for s in range(S):
            k = nl[s]
            sp = dict(thickness=thick[s, :k], density=dens[s, :k], temperature=temp[s, :k], microstructure=ms)
            if msl is not None:   # per-layer microstructure models and wetness
                sp = dict(thickness=thick[s, :k], frac_volume=dens[s, :k] / 916.7, temperature=temp[s, :k],
                          microstructure=[MS_NAMES[c] for c in msl[s, :k]], corr_length=p1[s, :k], radius=p1[s, :k],
                          stickiness=second[s, :k], repeat_distance=second[s, :k], liquid_water=lw[s, :k])
            elif ms == "exponential": sp["corr_length"] = p1[s, :k]
            else: sp["radius"] = p1[s, :k]; sp["stickiness"] = p2[s, :k]
            osub = None if sub is None else dict(kind="flat", eps=complex(sub[1][f, s], sub[2][f, s]), temperature=float(sub[3][s]))
            oatm = None if atm is None else dict(tb_down=atm[0][f], tb_up=atm[1][f], transmittance=atm[2][f])
            p = f * S + s
            try:
                det = {}
                ref = O.solve(sp, float(freqs[f]), theta, emmodel=em if eml is None else [FAMILY[c] for c in eml[s, :k]], mode=mode, theta_inc_deg=theta, n_max_stream=n, m_max=2,
                              method="schur_forcedtriu", substrate=osub, atmosphere=oatm, prune_deep_snowpack=prune, details=det,
                              process_coherent_layers_=with_coherent)
                n_pruned += bool(det["pruned_at"]) and min(det["pruned_at"]) < k
                n_coherent += with_coherent and len(det.get("kept_layers", range(k))) < k
            except O.OracleError as e:
                assert out.status[p] == e.status, (mode, n, em, out.status[p], e.status)
                n_refused += e.status == 6
                continue
            assert out.status[p] == 0, (mode, n, em, ms, f, s, out.status[p])
            n_checked += 1
            if mode == "P":
                if os.environ.get("STRESS_VERBOSE") and np.abs(out.values[p] - ref).max() > 1e-7:
                    print("  pair", p, "f", freqs[f], "err", np.abs(out.values[p] - ref).max(), "ms", None if msl is None else msl[s, :k], "lw", None if lw is None else lw[s, :k], "p1", p1[s, :k], "fv", dens[s, :k] / 916.7)
                worst_tb = max(worst_tb, float(np.abs(out.values[p] - ref).max()))
            else:
                sc = np.abs(ref[:2, :2]).max(axis=(0, 1))
                e_co = float((np.abs(out.values[p] - ref)[:2, :2] / sc).max())
                e_cx = float(np.abs(out.values[p][0, 1] / ref[0, 1] - 1).max())
                # the cross-pol own-scale error only means something where cross-pol is not vanishingly small
                ratio = float((np.abs(ref[0, 1]) / sc).min())
                if e_co > 1e-9 and os.environ.get("STRESS_DUMP"):
                    np.savez(os.path.join(ROOT, "gpurun_out", "stress_worst_%s_n%d_%d.npz" % (em, n, p)), nl=k, thickness=thick[s, :k], density=dens[s, :k],
                             temperature=temp[s, :k], p1=p1[s, :k], p2=(p2[s, :k] if p2 is not None else 0), freq=freqs[f], theta=theta,
                             sub=np.array([osub["eps"].real, osub["eps"].imag, osub["temperature"]]) if osub else np.zeros(0), gpu=out.values[p], ref=ref, em=em, ms=ms, n=n)
                if os.environ.get("STRESS_VERBOSE") and e_co > 1e-8:
                    print("  pair", p, "f", freqs[f], "e_co", e_co, "ms", None if msl is None else msl[s, :k], "lw", None if lw is None else lw[s, :k], "p1", p1[s, :k], "p2", None if p2 is None else p2[s, :k], "fv", dens[s, :k] / 916.7, "thick", thick[s, :k], "theta", theta, "sub", osub)
                    print("     gpu", out.values[p][:2, :2].ravel(), "ref", ref[:2, :2].ravel())
                    for meth in ("eig", "half_rank_eig"):
                        try:
                            alt = O.solve(sp, float(freqs[f]), theta, emmodel=em, mode=mode, theta_inc_deg=theta, n_max_stream=n, m_max=2,
                                          method=meth, substrate=osub, atmosphere=oatm, prune_deep_snowpack=prune)
                            print("     oracle %s vs default: co-pol scale %.2e" % (meth, float((np.abs(alt - ref)[:2, :2] / sc).max())))
                        except O.OracleError as e:
                            print("     oracle", meth, "fails:", e)
                if with_wetmicro and e_co > 1e-8:
                    # wet top layers absorb almost everything: backscatter at -70 dB and below is what the coherent
                    # subtraction leaves, defined only up to the spread of the reference's own eigensolvers -- the yardstick
                    # of tests/conftest.py:assert_backscatter_close (3 x that spread)
                    spread = 0.0
                    for meth in ("eig", "half_rank_eig"):
                        try:
                            alt = O.solve(sp, float(freqs[f]), theta, emmodel=em, mode=mode, theta_inc_deg=theta, n_max_stream=n, m_max=2,
                                          method=meth, substrate=osub, atmosphere=oatm, prune_deep_snowpack=prune)
                            spread = max(spread, float((np.abs(alt - ref)[:2, :2] / sc).max()))
                        except O.OracleError:
                            pass
                    assert e_co <= 3 * spread + 1e-8, (e_co, spread)
                    n_conditioned += 1
                    e_co = 0.0
                case_co = max(case_co, e_co); case_cx = max(case_cx, e_cx); case_ratio = min(case_ratio, ratio)
                worst_co = max(worst_co, e_co)
                if ratio > 1e-3: worst_cx = max(worst_cx, e_cx)
    extra = "" if mode == "P" else "  co %.1e  cross(own) %.1e  min cross/co %.1e" % (case_co, case_cx, case_ratio)
    print("%s n=%-3d %-22s %-20s sub=%d atm=%d  ok  (%.0f s)%s" % (mode, n, em, ms, sub is not None, atm is not None, time.time() - t0, extra), flush=True)
if with_coherent: print("process_coherent_layers: %d of the checked pairs lost at least one layer, %d pairs refused (status 6) by both" % (n_coherent, n_refused))
if with_wetmicro: print("wetmicro: microstructure model drawn per layer in the IBA cases, %d of the pairs with wet layers on top; %d active pairs beyond 1e-8 but within 3 x the spread of the oracle's own methods" % (n_wet, n_conditioned))
if with_family: print("family: emmodel drawn per layer among iba / iba_original / iba_maxwell_garnett in the IBA cases (the last two as SMRT_EM_IBA_HOST layers); %d pairs with such a layer on top" % n_family)
if with_sce: print("sce: symsce_torquato21 on the passive IBA / exponential cases (complex-wavenumber phase assembly), %d pairs" % n_family)
if with_prune: print("prune_deep_snowpack drawn per case: %d of the checked pairs were cut above their last layer" % n_pruned)
print("checked %d pairs: max |dTb| = %.2e K, backscatter max rel (co-pol scale) = %.2e, cross-pol own scale (where cross/co > 1e-3) = %.2e" % (n_checked, worst_tb, worst_co, worst_cx))
assert worst_tb < 1e-6 and worst_co < 1e-8 and worst_cx < 1e-6
