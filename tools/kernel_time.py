import os, sys, numpy as np
sys.path.insert(0, "/root/repo")
import bench
from smrt_amd import _native
if os.environ.get("SMRT_DORT_LIB"): _native.LIB_PATH = os.environ["SMRT_DORT_LIB"]
thick, dens, temp, lc = bench.synthetic_snowpacks(2)
b = _native.PackedBatch([20]*1024, thick, dens/916.7, temp, lc, None, bench.FREQS, np.deg2rad([55.0]))
ctx = _native.DortContext(0); ctx.upload(b)
for _ in range(3): ctx.launch()
ctx.sync(); ctx.total_kernel_ms(reset=True)
for _ in range(5): ctx.launch()
ctx.sync(); ms, n = ctx.total_kernel_ms()
out = ctx.download()
print(os.environ.get("SMRT_DORT_LIB", "default"), "kernel ms/launch %.2f  -> %.0f solves/s  failed %d" % (ms/n, 5120/(ms/n)*1e3, (out.status != 0).sum()))
