#!/usr/bin/env python
"""Per-stage shader-cycle breakdown of the finish kernels on one of bench.py's other configurations (needs a
-DSMRT_STAGE_TIMING build: python tools/build_variant.py timing -DSMRT_STAGE_TIMING):
   SMRT_DORT_LIB=smrt_amd/csrc/variants/libsmrt_dort_timing.so python tools/stage_profile_cfg.py [config: 2 | 3] [n_snowpacks]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from smrt_amd import _native  # noqa: E402

if os.environ.get("SMRT_DORT_LIB"):
    _native.LIB_PATH = os.environ["SMRT_DORT_LIB"]
config = int(sys.argv[1]) if len(sys.argv) > 1 else 2
S = int(sys.argv[2]) if len(sys.argv) > 2 else 64
batch, _, what = bench.make_workload(config, 0, False, S)
ctx = _native.DortContext(0)
ctx.upload(batch)
ctx.launch(); ctx.sync()
ctx.launch(); ctx.sync()
ms = ctx.last_kernel_ms()
st = ctx.stage_cycles()
sweeps = st.pop("_jacobi_sweeps")
sub = {k: st.pop(k) for k in ("_gj_panel", "_gj_update", "_gj_perm")}
print(what["what"])
print("gauss-jordan split (cycles/solve): panel %.0f  update %.0f  permutation %.0f" % tuple(
    sub[k] / batch.n_pairs for k in ("_gj_panel", "_gj_update", "_gj_perm")))
tot = sum(st.values())
print("pairs=%d kernel_ms=%.2f  solves/s=%.0f" % (batch.n_pairs, ms, batch.n_pairs / ms * 1e3))
for k, v in st.items():
    print("  %-11s %6.2f %%   %12.0f cycles/solve" % (k, 100 * v / max(tot, 1), v / batch.n_pairs))
