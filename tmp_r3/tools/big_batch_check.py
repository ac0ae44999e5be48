#!/usr/bin/env python
"""Chunked staging check: a batch larger than one staging pass (S snowpacks x 5 frequencies) must give exactly the
same numbers as the same pairs run in small batches."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from smrt_amd._native import DortContext, PackedBatch
S = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
thick, dens, temp, lc = bench.synthetic_snowpacks(5, S=S)
ctx = DortContext(0)
batch = PackedBatch([20] * S, thick, dens / 916.7, temp, lc, None, bench.FREQS, np.deg2rad([55.0]))
t0 = time.time(); big = ctx.run(batch); dt = time.time() - t0
print("big batch: %d pairs in %.2f s (%.0f solves/s incl. H2D/D2H), failed %d" % (batch.n_pairs, dt, batch.n_pairs / dt, int((big.status != 0).sum())))
idx = np.arange(0, S, max(1, S // 7))[:7]
small = PackedBatch([20] * len(idx), thick[idx], dens[idx] / 916.7, temp[idx], lc[idx], None, bench.FREQS, np.deg2rad([55.0]))
sm = ctx.run(small)
bv = big.values.reshape(5, S, 2, 1)[:, idx]
print("bitwise equal to small-batch results:", np.array_equal(bv, sm.values.reshape(5, len(idx), 2, 1)))
