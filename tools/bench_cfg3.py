#!/usr/bin/env python
"""Throughput of the 64-stream (global-workspace) kernel on a cfg3-like batch (DMRT-QCA-SR, 50 layers, 7 AMSR2
frequencies); S snowpacks (default 64 -> 448 solves)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smrt_amd import _native
if os.environ.get("SMRT_DORT_LIB"):
    _native.LIB_PATH = os.environ["SMRT_DORT_LIB"]
from smrt_amd._native import DortContext, PackedBatch
S = int(sys.argv[1]) if len(sys.argv) > 1 else 64
L = 50
rng = np.random.default_rng(3)
thick = np.concatenate([rng.uniform(0.05, 0.3, (S, L - 1)), np.full((S, 1), 100.0)], axis=1)
freqs = np.array([6.925e9, 7.3e9, 10.65e9, 18.7e9, 23.8e9, 36.5e9, 89e9])
batch = PackedBatch([L] * S, thick, rng.uniform(150, 450, (S, L)) / 916.7, rng.uniform(230, 270, (S, L)),
                    rng.uniform(5e-5, 1.5e-4, (S, L)), np.full((S, L), 0.2), freqs, np.deg2rad([55.0]),
                    emmodel="dmrt_qca_shortrange", microstructure="sticky_hard_spheres", n_max_stream=64)
ctx = DortContext(0)
ctx.upload(batch); ctx.launch(); ctx.sync(); ctx.launch(); ctx.sync()
ms = ctx.last_kernel_ms(); out = ctx.download()
print("cfg3-like: %d solves, kernel %.1f ms, %.0f solves/s, failed %d, sum N^3 = %.3e (68 N^3 -> %.2f TFLOP/s)" % (
    batch.n_pairs, ms, batch.n_pairs / ms * 1e3, int((out.status != 0).sum()), ctx.sum_n3(), 68 * ctx.sum_n3() / ms / 1e9))
if os.environ.get("SMRT_DORT_LIB"):  # profiling build: per-stage cycle shares of the fused kernel
    st = ctx.stage_cycles()
    tot = sum(v for k, v in st.items() if not k.startswith("_"))
    print("  ".join("%s %.1f%%" % (k, 100 * v / max(tot, 1)) for k, v in st.items() if not k.startswith("_")))
