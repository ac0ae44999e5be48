#!/usr/bin/env python
"""Where a block step of the strip kernels' elimination goes (build: python tools/build_variant.py stripinv -DSMRT_STRIP_TIMING_INV):
   SMRT_DORT_LIB=.../libsmrt_dort_stripinv.so python tools/strip_inv_profile.py [2 | 1] [n_snowpacks]"""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from smrt_amd import _native
if os.environ.get("SMRT_DORT_LIB"): _native.LIB_PATH = os.environ["SMRT_DORT_LIB"]
config = int(sys.argv[1]) if len(sys.argv) > 1 else 2
S = int(sys.argv[2]) if len(sys.argv) > 2 else (256 if config == 2 else 1024)
batch, _, what = bench.make_workload(config, 0, False, S)
ctx = _native.DortContext(0); ctx.upload(batch)
ctx.launch(); ctx.sync()
a = np.zeros(16)
ctx._check(ctx._lib.smrt_dort_stage_cycles(ctx._h, _native._dptr(a)), "stage_cycles")
L = int(batch.struct.n_layers_max)
per = batch.n_pairs * L * 3.0
names = {0: "wavefront 0: prologue (first 16 x 16 elimination + broadcast)", 1: "wavefront 0: its block-step work", 2: "wavefront 0: waiting at the step barrier",
         4: "next owner: diagonal tile update", 5: "next owner: elimination + interleaved updates", 6: "next owner: broadcast stores"}
print(what["what"])
for k, nm in names.items():
    print("  %-62s %9.0f ticks / inversion" % (nm, a[k] / per))
