#!/usr/bin/env python
"""Headline batch through the one-shot entry point smrt_dort_run (H2D of the packed inputs + three kernels + D2H of
results and diagnostics) next to the resident-input rate that bench.py reports.  DESIGN.md section 5."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from smrt_amd._native import DortContext, PackedBatch
thick, dens, temp, lc = bench.synthetic_snowpacks(2)
b = PackedBatch([20] * 1024, thick, dens / 916.7, temp, lc, None, bench.FREQS, np.deg2rad([55.0]))
ctx = DortContext(0)
ctx.run(b)
t0 = time.perf_counter()
for _ in range(5):
    out = ctx.run(b)
dt = (time.perf_counter() - t0) / 5
ctx.upload(b)
ctx.launch(); ctx.sync()
t0 = time.perf_counter()
for _ in range(5):
    ctx.launch()
ctx.sync()
dk = (time.perf_counter() - t0) / 5
print("one-shot run (H2D + kernels + D2H, host buffers): %.2f ms -> %.0f solves/s ; resident inputs: %.2f ms -> %.0f solves/s"
      % (dt * 1e3, b.n_pairs / dt, dk * 1e3, b.n_pairs / dk))
