"""Host-side cost of Model.run on the headline batch (1024 Snowpack objects x 5 channels): wall time of one run and a
cProfile of another -- what the plugin surface adds on top of the kernels.   python tools/model_run_profile.py"""
import cProfile, pstats, sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from smrt_amd import make_model, make_snowpack, sensor_list
S = 1024
thick, dens, temp, lc = bench.synthetic_snowpacks(2, S=S)
sps = [make_snowpack(thick[s], "exponential", density=dens[s], temperature=temp[s], corr_length=lc[s]) for s in range(S)]
sensor = sensor_list.passive(list(bench.FREQS), 55.0)
m = make_model("iba", "dort")
for _ in range(2): m.run(sensor, sps)
t0 = time.time(); m.run(sensor, sps); print("run: %.1f ms" % ((time.time() - t0) * 1e3))
pr = cProfile.Profile(); pr.enable(); m.run(sensor, sps); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(35)
