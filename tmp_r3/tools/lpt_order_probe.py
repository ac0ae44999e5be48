#!/usr/bin/env python
"""Does the dispatch order of the pairs matter?  The headline batch launched (a) in its natural order, (b) longest
estimated job first (sum of N_l^3 from smrt_dort_pair_cost, descending), (c) shortest first, (d) shuffled -- through
the pair-list entry point, which lets the caller choose the order of the workgroups.  Prints the pipeline time of each."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from smrt_amd._native import DortContext, PackedBatch

thick, dens, temp, lc = bench.synthetic_snowpacks(seed=2)
batch = PackedBatch([bench.N_LAYERS] * bench.N_SNOWPACKS, thick, dens / 916.7, temp, lc, None, bench.FREQS,
                    np.deg2rad([bench.THETA_DEG]), emmodel="iba", microstructure="exponential", mode="P",
                    n_max_stream=bench.N_STREAMS)
ctx = DortContext(0)
ctx.upload(batch)
cost = ctx.pair_cost()
print("pair cost (sum N^3): min %.3g  median %.3g  max %.3g" % (cost.min(), np.median(cost), cost.max()))
n = batch.n_pairs
orders = {"natural": np.arange(n), "longest first": np.argsort(-cost, kind="stable"), "shortest first": np.argsort(cost, kind="stable"),
          "shuffled": np.random.default_rng(0).permutation(n)}
ref = None
for name, order in orders.items():
    ctx.upload(batch, pairs=order)
    ts = []
    for _ in range(6):
        ctx.launch(); ctx.sync(); ts.append(ctx.last_kernel_ms())
    out = ctx.download()
    vals = np.empty_like(out.values); vals[order] = out.values
    if ref is None: ref = vals
    print("%-15s %.2f ms (min of 6: %.2f)  same results: %s" % (name, np.median(ts[1:]), min(ts), np.array_equal(vals, ref)))
