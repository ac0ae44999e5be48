#!/usr/bin/env python
"""cProfile of Model.run on the headline batch (1024 Snowpack objects x 5 channels): where the host time of the plugin
surface goes on top of the device time."""
import cProfile, os, pstats, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from smrt_amd import make_model, make_snowpack
from smrt_amd.core.sensor import passive

thick, dens, temp, lc = bench.synthetic_snowpacks(seed=2)
sps = [make_snowpack(thick[s], "exponential", density=dens[s], temperature=temp[s], corr_length=lc[s]) for s in range(len(thick))]
sensor = passive(list(bench.FREQS), bench.THETA_DEG)
m = make_model("iba", "dort", rtsolver_options=dict(n_max_stream=bench.N_STREAMS))
m.run(sensor, sps); m.run(sensor, sps)
t0 = time.time(); m.run(sensor, sps); print("Model.run: %.2f ms" % ((time.time() - t0) * 1e3))
pr = cProfile.Profile(); pr.enable(); m.run(sensor, sps); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
