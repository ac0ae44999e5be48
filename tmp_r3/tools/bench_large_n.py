#!/usr/bin/env python
"""Passive batch at a large stream count (N = 2 x streams rows: 128 streams -> N = 256, 192 -> N = 384), thin-layer laws
of BASELINE configs[3]: time, solves/s, TFLOP/s and -- with a -DSMRT_STAGE_TIMING build in SMRT_DORT_LIB -- the share of
every stage.   python tools/bench_large_n.py [streams] [snowpacks] [layers]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smrt_amd._native import DortContext, PackedBatch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
S = int(sys.argv[2]) if len(sys.argv) > 2 else 256
L = int(sys.argv[3]) if len(sys.argv) > 3 else 6
rng = np.random.default_rng(4)
thick = np.concatenate([rng.uniform(0.02, 0.10, (S, L - 1)), np.full((S, 1), 1000.0)], axis=1)
dens, temp, lc = rng.uniform(150, 450, (S, L)), rng.uniform(230, 270, (S, L)), rng.uniform(5e-5, 3e-4, (S, L))
b = PackedBatch([L] * S, thick, dens / 916.7, temp, lc, None, [5.405e9], np.deg2rad([40.0]), emmodel="iba",
                microstructure="exponential", mode="P", n_max_stream=n)
ctx = DortContext(0)
ctx.upload(b); ctx.launch(); ctx.sync(); ctx.launch(); ctx.sync()
ms = ctx.last_kernel_ms(); out = ctx.download()
print("passive, %d streams, %d layers, %d pairs: kernel %.1f ms, %.1f solves/s, failed %d, 68 sum N^3 -> %.2f TFLOP/s" % (
    n, L, S, ms, S / ms * 1e3, int((out.status != 0).sum()), 68 * ctx.sum_n3() / ms / 1e9))
if os.environ.get("SMRT_DORT_LIB"):
    st = ctx.stage_cycles()
    tot = sum(v for k, v in st.items() if not k.startswith("_"))
    print("  ".join("%s %.1f%%" % (k, 100 * v / max(tot, 1)) for k, v in st.items() if not k.startswith("_")))
    print("jacobi sweeps per layer", st["_jacobi_sweeps"] / S / L)
