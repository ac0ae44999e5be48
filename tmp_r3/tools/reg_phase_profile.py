#!/usr/bin/env python
"""Shader-clock breakdown of the register-resident finish kernel by phase on the headline batch (needs a build with
-DSMRT_REG_TIMING:  python tools/build_variant.py regtiming -DSMRT_REG_TIMING
   SMRT_DORT_LIB=smrt_amd/csrc/variants/libsmrt_dort_regtiming.so python tools/reg_phase_profile.py [n_snowpacks])."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from smrt_amd import _native
if os.environ.get("SMRT_DORT_LIB"): _native.LIB_PATH = os.environ["SMRT_DORT_LIB"]
S = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
thick, dens, temp, lc = bench.synthetic_snowpacks(2, S=S)
b = _native.PackedBatch([20] * S, thick, dens / 916.7, temp, lc, None, bench.FREQS, np.deg2rad([55.0]))
ctx = _native.DortContext(0); ctx.set_pipeline(3); ctx.upload(b)
ctx.launch(); ctx.sync(); ctx.total_kernel_ms(reset=True)
ctx.launch(); ctx.sync(); ms, n = ctx.total_kernel_ms()
a = np.zeros(16)
ctx._check(ctx._lib.smrt_dort_stage_cycles(ctx._h, _native._dptr(a)), "stage_cycles")
names = ["setup", "vectors", "B' + At", "A+", "H^T", "invert (x3)", "stage 0/1 prep", "y, T2", "C'", "interface / surface prep", "post (Z, C_u)", "surface"]
tot = a[:12].sum()
print("kernel ms %.2f (all three kernels); shader-clock counts per pair-layer (%d pairs x 20 layers):" % (ms / n, 5 * S))
for k, nm in enumerate(names):
    print("  %-26s %6.2f %%  %9.0f ticks / layer" % (nm, 100 * a[k] / tot, a[k] / (5 * S * 20)))
print("  total %.0f ticks / layer" % (tot / (5 * S * 20)))
