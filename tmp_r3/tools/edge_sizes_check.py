#!/usr/bin/env python
"""Sizing edge cases on the GPU box against the CPU oracle: many layers (60, 120), many viewing / incidence angles,
ragged layer counts, LDS and global-workspace paths.  Not part of the test-suite (oracle time)."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import dort_oracle as O
from smrt_amd._native import DortContext, PackedBatch
rng = np.random.default_rng(9)
ctx = DortContext(0)
for (L, n, ntheta, mode) in [(60, 32, 8, "P"), (120, 16, 3, "P"), (40, 16, 6, "A"), (25, 64, 2, "P")]:
    S = 2
    thick = np.concatenate([rng.uniform(0.02, 0.2, (S, L - 1)), np.full((S, 1), 50.0)], axis=1)
    dens, temp, lc = rng.uniform(150, 450, (S, L)), rng.uniform(230, 270, (S, L)), rng.uniform(5e-5, 3e-4, (S, L))
    theta = np.linspace(10, 65, ntheta)
    b = PackedBatch([L, L - 3], thick, dens / 916.7, temp, lc, None, [18.7e9], np.deg2rad(theta), emmodel="iba", microstructure="exponential", mode=mode, n_max_stream=n, m_max=2)
    out = ctx.run(b)
    worst = 0
    for s, k in enumerate([L, L - 3]):
        sp = dict(thickness=thick[s, :k], density=dens[s, :k], temperature=temp[s, :k], microstructure="exponential", corr_length=lc[s, :k])
        ref = O.solve(sp, 18.7e9, theta, mode=mode, theta_inc_deg=theta, n_max_stream=n, m_max=2, method="schur_forcedtriu")
        if mode == "P": worst = max(worst, np.abs(out.values[s] - ref).max())
        else: worst = max(worst, (np.abs(out.values[s] - ref)[:2, :2] / np.abs(ref[:2, :2]).max(axis=(0, 1))).max())
    print("L=%d n=%d ntheta=%d mode=%s status=%s worst=%.2e" % (L, n, ntheta, mode, out.status, worst))
