#!/usr/bin/env python
"""Throughput of the active-mode kernel on a reduced config-4 batch (BASELINE configs[3] laws: IBA, Sentinel-1 C band,
30 thin layers over a deep one, incidence 20..45 deg, m_max = 2) at 16 streams (N = 48, LDS-resident kernel) and at
32 streams (N = 96, global-workspace kernel).  S snowpacks (default 512)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smrt_amd._native import DortContext, PackedBatch
S = int(sys.argv[1]) if len(sys.argv) > 1 else 512
L = 30
rng = np.random.default_rng(4)
thick = np.concatenate([rng.uniform(0.02, 0.10, (S, L - 1)), np.full((S, 1), 1000.0)], axis=1)
dens, temp, lc = rng.uniform(150, 450, (S, L)), rng.uniform(230, 270, (S, L)), rng.uniform(5e-5, 3e-4, (S, L))
theta = np.deg2rad(np.arange(20.0, 46.0, 5.0))
ctx = DortContext(0)
for n, s_ in ((16, S), (32, max(S // 8, 32))):
    batch = PackedBatch([L] * s_, thick[:s_], dens[:s_] / 916.7, temp[:s_], lc[:s_], None, [5.405e9], theta, emmodel="iba",
                        microstructure="exponential", mode="A", n_max_stream=n, m_max=2)
    ctx.upload(batch); ctx.launch(); ctx.sync(); ctx.launch(); ctx.sync()
    ms = ctx.last_kernel_ms(); out = ctx.download()
    print("active, %d streams (N = %d): %d solves, kernel %.1f ms, %.0f solves/s, failed %d, 68 sum N^3 -> %.2f TFLOP/s; sigma0_VV[0] = %s dB"
          % (n, 3 * n, batch.n_pairs, ms, batch.n_pairs / ms * 1e3, int((out.status != 0).sum()), 68 * ctx.sum_n3() / ms / 1e9,
             np.round(10 * np.log10(4 * np.pi * np.cos(theta) * out.values[0, 0, 0]), 3)))
