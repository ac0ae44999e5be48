#!/usr/bin/env python
"""BASELINE configs[2] shape on one GPU (DMRT-QCA-SR + DORT passive, 50 layers, 64 streams, 7 AMSR2 frequencies):
S snowpacks (default 64 -> 448 solves).  Prints ONE JSON line shaped like bench.py's: value (resident-input rate),
roofline with the ACTUAL sum of N_l^3 (smrt_dort_sum_n3: total reflection removes streams layer by layer), failed solves.
Parity of this shape at batch scale is a pytest matter (tests/test_gpu_parity.py::test_cfg3_shape_batch_*).
   python tools/bench_cfg3.py [n_snowpacks] [steps]"""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smrt_amd import _native
if os.environ.get("SMRT_DORT_LIB"):
    _native.LIB_PATH = os.environ["SMRT_DORT_LIB"]
from smrt_amd._native import DortContext, PackedBatch
S = int(sys.argv[1]) if len(sys.argv) > 1 else 64
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
L = 50
rng = np.random.default_rng(3)
thick = np.concatenate([rng.uniform(0.05, 0.3, (S, L - 1)), np.full((S, 1), 100.0)], axis=1)
freqs = np.array([6.925e9, 7.3e9, 10.65e9, 18.7e9, 23.8e9, 36.5e9, 89e9])
batch = PackedBatch([L] * S, thick, rng.uniform(150, 450, (S, L)) / 916.7, rng.uniform(230, 270, (S, L)),
                    rng.uniform(5e-5, 1.5e-4, (S, L)), np.full((S, L), 0.2), freqs, np.deg2rad([55.0]),
                    emmodel="dmrt_qca_shortrange", microstructure="sticky_hard_spheres", n_max_stream=64)
ctx = DortContext(0)
ctx.upload(batch); ctx.launch(); ctx.sync()
ctx.total_kernel_ms(reset=True)
for _ in range(steps): ctx.launch()
ctx.sync()
ms_tot, n = ctx.total_kernel_ms(); ms = ms_tot / n
out = ctx.download()
flops = 68.0 * ctx.sum_n3()
ach = flops / (ms * 1e-3) / 1e12
print(json.dumps({
    "metric": "snowpack x frequency DORT solves/sec (50 layers, 64 streams)", "value": batch.n_pairs / ms * 1e3,
    "unit": "solves/s", "n_gpus": 1, "steps": steps, "ms_per_step": ms, "dtype": "f64", "data": "synthetic",
    "config": {"workload": "BASELINE configs[2] shape: DMRT-QCA-SR + DORT passive, 50 layers, 64 streams, 7 AMSR2 "
                           "frequencies, %d snowpacks = %d solves per step, inputs resident" % (S, batch.n_pairs),
               "failed_solves": int((out.status != 0).sum())},
    "roofline": {"bound": "mfma", "achieved": ach, "peak": 78.6, "unit": "TFLOP/s", "frac": ach / 78.6, "traffic": None,
                 "flops_per_launch": flops, "kernel_ms": ms,
                 "kernel": "prep_gmem + jacobi<512> + finish_gmem<512> (64 < N <= 128 pipeline), summed HIP-event time"}}))
if os.environ.get("SMRT_DORT_LIB") and os.environ.get("SMRT_STAGES"):  # profiling build: per-stage cycle shares
    st = ctx.stage_cycles()
    tot = sum(v for k, v in st.items() if not k.startswith("_"))
    print("  ".join("%s %.1f%%" % (k, 100 * v / max(tot, 1)) for k, v in st.items() if not k.startswith("_")))
