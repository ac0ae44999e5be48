#!/usr/bin/env python
"""Active mode (backscatter) on deliberately hard inputs, every pair against the CPU oracle (reference default method):
1.26 ... 94 GHz, layers from 0.1 mm to 1000 m, ice volume fractions 0.05 ... 0.49, correlation lengths up to the
renormalisation limit, 4 ... 30 streams (N = 12 ... 90: LDS and global-workspace pipelines), with and without a flat
substrate.  Errors relative to the co-polarised scale of the pair; cross-polarised terms also on their own scale where
they are not vanishing.  Pairs the oracle refuses must come back with the same status.
    python tools/stress_active_extremes.py [seed] [n_cases]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import dort_oracle as O  # checker only
from smrt_amd._native import DortContext, PackedBatch

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 8
rng = np.random.default_rng(seed)
ctx = DortContext(0)
worst_co = worst_cx = 0.0
checked = refused = mism = n_beyond = 0
for case in range(n_cases):
    S, L = 4, int(rng.integers(1, 7))
    n_str = int(rng.choice([4, 8, 12, 16, 21, 30]))
    thick = 10.0 ** rng.uniform(-4, 0.5, (S, L)); thick[:, -1] = rng.choice([0.5, 1000.0], S)
    fv = rng.uniform(0.05, 0.49, (S, L)); temp = rng.uniform(200, 272.9, (S, L))
    lc = 10.0 ** rng.uniform(-5, -3.2, (S, L))
    freqs = np.sort(rng.choice([1.26e9, 5.4e9, 13.4e9, 35e9, 94e9], 2, replace=False))
    theta = np.sort(rng.uniform(5, 65, 2))
    sub = None
    if rng.random() < 0.5:
        sub = ("flat", np.full((2, S), rng.uniform(2, 30)), np.full((2, S), rng.uniform(0.01, 5)), rng.uniform(240, 273, S))
    b = PackedBatch([L] * S, thick, fv, temp, lc, None, freqs, np.deg2rad(theta), mode="A", n_max_stream=n_str, m_max=2, substrate=sub)
    out = ctx.run(b)
    for fi, f in enumerate(freqs):
        for s in range(S):
            p = fi * S + s
            sp = dict(thickness=thick[s], frac_volume=fv[s], temperature=temp[s], microstructure="exponential", corr_length=lc[s])
            osub = None if sub is None else dict(kind="flat", eps=complex(sub[1][fi, s], sub[2][fi, s]), temperature=float(sub[3][s]))
            try:
                ref = O.solve(sp, float(f), theta, mode="A", theta_inc_deg=theta, n_max_stream=n_str, m_max=2, method="schur_forcedtriu", substrate=osub)
                st = 0
            except O.OracleError as e:
                st = e.status
            if st != out.status[p]:
                mism += 1
                print("status mismatch: case %d pair %d oracle %d device %d (n = %d, f = %.2f GHz)" % (case, p, st, out.status[p], n_str, f / 1e9))
                continue
            if st != 0:
                refused += 1
                continue
            checked += 1
            sc = np.abs(ref[:2, :2]).max(axis=(0, 1))
            e_co = float((np.abs(out.values[p] - ref)[:2, :2] / sc).max())
            ratio = float((np.abs(ref[0, 1]) / sc).min())
            e_cx = float(np.abs(out.values[p][0, 1] / ref[0, 1] - 1).max()) if ratio > 1e-3 else 0.0
            if e_co > 1e-8 or e_cx > 1e-6:
                # the yardstick of the tests: how far the reference's OTHER diagonalisation methods are from its default
                sp_co = sp_cx = 0.0
                for meth in ("eig", "half_rank_eig"):
                    try:
                        alt = O.solve(sp, float(f), theta, mode="A", theta_inc_deg=theta, n_max_stream=n_str, m_max=2, method=meth, substrate=osub)
                    except O.OracleError:
                        continue
                    sp_co = max(sp_co, float((np.abs(alt - ref)[:2, :2] / sc).max()))
                    if ratio > 1e-3: sp_cx = max(sp_cx, float(np.abs(alt[0, 1] / ref[0, 1] - 1).max()))
                beyond = (e_co > max(1e-8, 3 * sp_co)) or (e_cx > max(1e-6, 3 * sp_cx))
                n_beyond += beyond
                print("case %d pair %d: co %.2e (oracle's other methods: %.2e) cross(own) %.2e (%.2e)  L = %d, n = %d, f = %.2f GHz, thinnest %.2e m, sub %d%s" % (
                    case, p, e_co, sp_co, e_cx, sp_cx, L, n_str, f / 1e9, thick[s].min(), sub is not None, "  BEYOND 3 x SPREAD" if beyond else ""))
            worst_co = max(worst_co, e_co); worst_cx = max(worst_cx, e_cx)
print("seed %d: %d pairs checked, backscatter max rel (co-pol scale) = %.2e, cross-pol own scale (where cross/co > 1e-3) = %.2e; %d refused by "
      "both; %d status mismatches; %d pairs beyond 1e-8 AND beyond 3 x the spread of the oracle's own methods" % (seed, checked, worst_co, worst_cx, refused, mism, n_beyond))
