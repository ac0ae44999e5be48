#!/usr/bin/env python
"""Per-stage shader-cycle breakdown of the pair kernel on the bench workload (needs a -DSMRT_STAGE_TIMING build:
   hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DSMRT_STAGE_TIMING -o /tmp/libsmrt_dort_prof.so smrt_amd/csrc/dort_hip.hip
   SMRT_DORT_LIB=/tmp/libsmrt_dort_prof.so python tools/stage_profile.py [threads] [n_snowpacks])."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from smrt_amd import _native  # noqa: E402

if os.environ.get("SMRT_DORT_LIB"):
    _native.LIB_PATH = os.environ["SMRT_DORT_LIB"]
threads = int(sys.argv[1]) if len(sys.argv) > 1 else 256
S = int(sys.argv[2]) if len(sys.argv) > 2 else 256
thick, dens, temp, lc = bench.synthetic_snowpacks(2, S=S)
batch = _native.PackedBatch([bench.N_LAYERS] * S, thick, dens / 916.7, temp, lc, None, bench.FREQS,
                            np.deg2rad([55.0]))
ctx = _native.DortContext(0)
ctx.set_block_threads(threads)
ctx.upload(batch)
ctx.launch(); ctx.sync()
ctx.launch(); ctx.sync()
ms = ctx.last_kernel_ms()
st = ctx.stage_cycles()
sweeps = st.pop("_jacobi_sweeps")
sub = {k: st.pop(k) for k in ("_gj_panel", "_gj_update", "_gj_perm")}
print("gauss-jordan split (cycles/solve): panel %.0f  update %.0f  permutation %.0f" % tuple(
    sub[k] / batch.n_pairs for k in ("_gj_panel", "_gj_update", "_gj_perm")))
print("jacobi sweeps per layer-problem: %.2f" % (sweeps / batch.n_pairs / bench.N_LAYERS))
tot = sum(st.values())
print("threads=%d pairs=%d kernel_ms=%.2f  solves/s=%.0f" % (threads, batch.n_pairs, ms, batch.n_pairs / ms * 1e3))
for k, v in st.items():
    print("  %-11s %6.2f %%   %10.0f cycles/solve" % (k, 100 * v / max(tot, 1), v / batch.n_pairs))
