#!/usr/bin/env python
"""Kernel time of the headline batch (HIP events) for a given library build / pipeline shape:
   [SMRT_DORT_LIB=path] python tools/kernel_time.py [pipeline: 1 | 2 | 0]"""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from smrt_amd import _native
if os.environ.get("SMRT_DORT_LIB"): _native.LIB_PATH = os.environ["SMRT_DORT_LIB"]
pipe = int(sys.argv[1]) if len(sys.argv) > 1 else 1
thick, dens, temp, lc = bench.synthetic_snowpacks(2)
b = _native.PackedBatch([20]*1024, thick, dens/916.7, temp, lc, None, bench.FREQS, np.deg2rad([55.0]))
ctx = _native.DortContext(0); ctx.set_pipeline(pipe); ctx.upload(b)
for _ in range(3): ctx.launch()
ctx.sync(); ctx.total_kernel_ms(reset=True)
for _ in range(5): ctx.launch()
ctx.sync(); ms, n = ctx.total_kernel_ms()
out = ctx.download()
print(os.environ.get("SMRT_DORT_LIB", "default"), "pipeline", pipe, "kernel ms/launch %.2f  -> %.0f solves/s  failed %d  Tb[0]=%s" % (ms/n, 5120/(ms/n)*1e3, (out.status != 0).sum(), out.values[0].ravel()))
