#!/usr/bin/env python
"""Shader-clock breakdown of the strip finish kernels by phase (needs a build with -DSMRT_STRIP_TIMING:
   python tools/build_variant.py striptiming -DSMRT_STRIP_TIMING
   SMRT_DORT_LIB=smrt_amd/csrc/variants/libsmrt_dort_striptiming.so python tools/strip_phase_profile.py [2 | 1] [n_snowpacks]
 2: the configs[2] shape (eight wavefronts, N <= 128); 1: the headline batch with SMRT_DORT_FINISH_STRIP4=1 (four wavefronts)."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from smrt_amd import _native
if os.environ.get("SMRT_DORT_LIB"): _native.LIB_PATH = os.environ["SMRT_DORT_LIB"]
config = int(sys.argv[1]) if len(sys.argv) > 1 else 2
S = int(sys.argv[2]) if len(sys.argv) > 2 else (256 if config == 2 else 1024)
batch, _, what = bench.make_workload(config, 0, False, S)
ctx = _native.DortContext(0); ctx.upload(batch)
ctx.launch(); ctx.sync(); ctx.total_kernel_ms(reset=True)
ctx.launch(); ctx.sync(); ms, n = ctx.total_kernel_ms()
a = np.zeros(16)
ctx._check(ctx._lib.smrt_dort_stage_cycles(ctx._h, _native._dptr(a)), "stage_cycles")
names = ["setup / output", "loads: L+ -> LDS, vectors", "W = L+ B', A+ = L+^-T B', r", "W^T -> At, park, C^ -> LDS", "T1 = C^^T A+", "H^T = A+^T T1 (+ put A+)",
         "inversion 1 (P)", "inversion 2 (M3)", "Theta -> LDS, At back, T2", "C^' = At^T T2, sums, c'", "interface coefficients, Y, Nn", "inversion 3 (Y)", "Z = Nn Y^-1, C_u, sums", "surface"]
L = int(batch.struct.n_layers_max)
tot = a[:14].sum()
print(what["what"])
print("kernel ms %.2f (all three kernels); shader-clock counts of thread 0 per pair-layer (%d pairs x %d layers):" % (ms / n, batch.n_pairs, L))
for k, nm in enumerate(names):
    print("  %-34s %6.2f %%  %9.0f ticks / layer" % (nm, 100 * a[k] / tot, a[k] / (batch.n_pairs * L)))
print("  total %.0f ticks / layer" % (tot / (batch.n_pairs * L)))
