#!/usr/bin/env python
"""Register-resident finish kernel (set_pipeline(3)) against the two-slot finish kernel (pipeline 1) on the headline
batch: maximum difference of the brightness temperatures, statuses, kernel time of both.
   python tools/finish_reg_check.py [n_snowpacks]"""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from smrt_amd import _native
S = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
thick, dens, temp, lc = bench.synthetic_snowpacks(2, S=S)
b = _native.PackedBatch([20] * S, thick, dens / 916.7, temp, lc, None, bench.FREQS, np.deg2rad([55.0]))
res = {}
for pipe in (4, 3):   # 4: never the register-resident finish kernel, 3: wherever it is supported
    ctx = _native.DortContext(0); ctx.set_pipeline(pipe); ctx.upload(b)
    for _ in range(2): ctx.launch()
    ctx.sync(); ctx.total_kernel_ms(reset=True)
    for _ in range(5): ctx.launch()
    ctx.sync(); ms, n = ctx.total_kernel_ms()
    out = ctx.download()
    res[pipe] = out
    print("pipeline", pipe, "kernel ms/launch %.2f -> %.0f solves/s, failed %d, Tb[0] = %s" % (
        ms / n, 5 * S / (ms / n) * 1e3, (out.status != 0).sum(), out.values[0].ravel()), flush=True)
    ctx.close()
ok = (res[4].status == 0) & (res[3].status == 0)
print("max |Tb(reg) - Tb(two-slot)| = %.3e K over %d pairs" % (np.abs(res[4].values[ok] - res[3].values[ok]).max(), ok.sum()))
