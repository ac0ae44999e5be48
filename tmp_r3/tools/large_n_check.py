#!/usr/bin/env python
"""Stream counts beyond the pipelines (streams x polarisations > 128: the fused global-workspace kernel with scalar
dense steps) against the CPU oracle, with timings.  Usage: python tools/large_n_check.py [case ...] with cases like
P100 (passive, 100 streams) or A50 (active, 50 streams -> N = 150)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import dort_oracle as O
from smrt_amd._native import DortContext, PackedBatch
cases = sys.argv[1:] or ["P100", "P128", "A50", "A85"]
rng = np.random.default_rng(17)
ctx = DortContext(0)
for case in cases:
    mode, n = case[0], int(case[1:])
    S, L = 2, 3
    thick = np.concatenate([rng.uniform(0.05, 0.3, (S, L - 1)), np.full((S, 1), 20.0)], axis=1)
    dens, temp, lc = rng.uniform(150, 450, (S, L)), rng.uniform(230, 270, (S, L)), rng.uniform(5e-5, 3e-4, (S, L))
    theta = np.array([30.0, 50.0])
    freq = 13.4e9 if mode == "A" else 36.5e9
    b = PackedBatch([L] * S, thick, dens / 916.7, temp, lc, None, [freq], np.deg2rad(theta), emmodel="iba",
                    microstructure="exponential", mode=mode, n_max_stream=n, m_max=2)
    t0 = time.time(); out = ctx.run(b); dt = time.time() - t0
    worst = 0.0
    t1 = time.time()
    for s in range(S):
        sp = dict(thickness=thick[s], density=dens[s], temperature=temp[s], microstructure="exponential", corr_length=lc[s])
        ref = O.solve(sp, freq, theta, mode=mode, theta_inc_deg=theta, n_max_stream=n, m_max=2, method="schur_forcedtriu")
        if mode == "P": worst = max(worst, np.abs(out.values[s] - ref).max())
        else: worst = max(worst, (np.abs(out.values[s] - ref)[:2, :2] / np.abs(ref[:2, :2]).max(axis=(0, 1))).max())
    print("%s: N=%d status=%s gpu %.2f s (2 pairs x %d layers), oracle %.1f s, worst %s %.2e" % (
        case, n * (3 if mode == "A" else 2), list(out.status), dt, L, time.time() - t1, "K" if mode == "P" else "rel", worst), flush=True)
