#!/usr/bin/env python
"""BASELINE configs[3] shape on one GPU: IBA + DORT active, Sentinel-1 C band, 30 layers, 128 streams (N = 384 for the
azimuth modes m >= 1), m_max = 2.  Runs on the fused global-workspace kernel with scalar dense steps.
Usage: python tools/bench_cfg4.py [n_snowpacks]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smrt_amd._native import DortContext, PackedBatch
S = int(sys.argv[1]) if len(sys.argv) > 1 else 256
L = 30
rng = np.random.default_rng(4)
thick = np.concatenate([rng.uniform(0.02, 0.10, (S, L - 1)), np.full((S, 1), 1000.0)], axis=1)
dens, temp, lc = rng.uniform(150, 450, (S, L)), rng.uniform(230, 270, (S, L)), rng.uniform(5e-5, 3e-4, (S, L))
theta = np.arange(20.0, 46.0, 5.0)
b = PackedBatch([L] * S, thick, dens / 916.7, temp, lc, None, [5.405e9], np.deg2rad(theta), emmodel="iba",
                microstructure="exponential", mode="A", n_max_stream=128, m_max=2)
ctx = DortContext(0)
ctx.upload(b)
t0 = time.time(); ctx.launch(); ctx.sync(); dt = time.time() - t0
out = ctx.download()
ok = int((out.status == 0).sum())
vv = 10 * np.log10(4 * np.pi * np.cos(np.deg2rad(theta)) * out.values[0][0, 0])
print("cfg4 shape: %d pairs (30 layers, 128 streams, m_max 2) in %.1f s = %.1f solves/s, ok %d/%d, kernel %.1f s" % (
    S, dt, S / dt, ok, S, ctx.last_kernel_ms() / 1e3))
print("sigma0_VV(dB) of pair 0 at 20..45 deg:", np.round(vv, 3))
