#!/usr/bin/env python
"""BASELINE configs[3] shape on one GPU: IBA + DORT active, Sentinel-1 C band, 30 layers, 128 streams (N = 256 for mode 0,
384 for the azimuth modes m >= 1), m_max = 2, on the three-kernel pipeline for N > 128.  Prints ONE JSON
line shaped like bench.py's: value, roofline with the ACTUAL sum over pairs, modes and layers of N_l^3, failed solves,
and sigma0_VV of the first pair.  Parity of this shape: tests/test_gpu_parity.py::test_active_full_size_cfg4_shape
(reference fixtures) and ::test_cfg4_shape_batch_through_staging_chunks.
   python tools/bench_cfg4.py [n_snowpacks]"""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smrt_amd._native import DortContext, PackedBatch
S = int(sys.argv[1]) if len(sys.argv) > 1 else 256
L = 30
rng = np.random.default_rng(4)
thick = np.concatenate([rng.uniform(0.02, 0.10, (S, L - 1)), np.full((S, 1), 1000.0)], axis=1)
dens, temp, lc = rng.uniform(150, 450, (S, L)), rng.uniform(230, 270, (S, L)), rng.uniform(5e-5, 3e-4, (S, L))
theta = np.arange(20.0, 46.0, 5.0)
b = PackedBatch([L] * S, thick, dens / 916.7, temp, lc, None, [5.405e9], np.deg2rad(theta), emmodel="iba",
                microstructure="exponential", mode="A", n_max_stream=128, m_max=2)
ctx = DortContext(0)
ctx.upload(b)
ctx.launch(); ctx.sync()
ms = ctx.last_kernel_ms()
out = ctx.download()
flops = 68.0 * ctx.sum_n3()
ach = flops / (ms * 1e-3) / 1e12
vv = 10 * np.log10(4 * np.pi * np.cos(np.deg2rad(theta)) * out.values[0][0, 0])
print(json.dumps({
    "metric": "snowpack x frequency DORT solves/sec (active, 30 layers, 128 streams, m_max 2)", "value": S / ms * 1e3,
    "unit": "solves/s", "n_gpus": 1, "steps": 1, "ms_per_step": ms, "dtype": "f64", "data": "synthetic",
    "config": {"workload": "BASELINE configs[3] shape: IBA + DORT active, Sentinel-1 (5.405 GHz, 20..45 deg), 30 layers, "
                           "128 streams, m_max 2, %d snowpacks, inputs resident" % S,
               "failed_solves": int((out.status != 0).sum()), "sigma0_VV_dB_pair0": [round(float(v), 4) for v in vv]},
    "roofline": {"bound": "mfma", "achieved": ach, "peak": 78.6, "unit": "TFLOP/s", "frac": ach / 78.6, "traffic": None,
                 "flops_per_launch": flops, "kernel_ms": ms,
                 "kernel": "active_big prep + jacobi_big + active_big finish (128 < N <= 384 pipeline), summed HIP-event time"}}))
