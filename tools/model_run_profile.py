"""Host-side cost of Model.run on the headline batch (S Snowpack objects x 5 channels, default 1024): wall time of one
run, the same packed batch through the C entry point (run_on_devices: H2D + kernels + D2H, one host thread per listed
device), their difference = what the plugin surface adds on the host, and a cProfile of another run.
    python tools/model_run_profile.py [S] [n_devices: GPU 0 listed that many times, as VERDICT r4 item 5 asks]"""
import cProfile, pstats, sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from smrt_amd import make_model, make_snowpack, sensor_list
from smrt_amd._native import PackedBatch
from smrt_amd.rtsolver.dort import run_on_devices
from smrt_amd.runner.hip_batch_runner import HipBatchRunner
S = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
ndev = int(sys.argv[2]) if len(sys.argv) > 2 else 1
devices = [0] * ndev
thick, dens, temp, lc = bench.synthetic_snowpacks(2, S=S)
sps = [make_snowpack(thick[s], "exponential", density=dens[s], temperature=temp[s], corr_length=lc[s]) for s in range(S)]
sensor = sensor_list.passive(list(bench.FREQS), 55.0)
m = make_model("iba", "dort")
runner = HipBatchRunner(devices=devices)
for _ in range(2): m.run(sensor, sps, runner=runner)
ts = []
for _ in range(5):
    t0 = time.time(); m.run(sensor, sps, runner=runner); ts.append((time.time() - t0) * 1e3)
L = thick.shape[1]
batch = PackedBatch([L] * S, thick, dens / 916.7, temp, lc, None, list(bench.FREQS), np.deg2rad([55.0]), emmodel="iba",
                    microstructure="exponential", mode="P", n_max_stream=bench.N_STREAMS)
cost = np.ones(batch.n_pairs)
for _ in range(2): run_on_devices(batch, devices, cost=cost)
tc = []
for _ in range(5):
    t0 = time.time(); run_on_devices(batch, devices, cost=cost); tc.append((time.time() - t0) * 1e3)
run_ms, c_ms = float(np.median(ts)), float(np.median(tc))
print("S = %d snowpacks x %d channels on devices %s: Model.run %.1f ms, C entry point on the packed batch %.1f ms, "
      "host side of Model.run = %.1f ms (medians of 5)" % (S, len(bench.FREQS), devices, run_ms, c_ms, run_ms - c_ms))
pr = cProfile.Profile(); pr.enable(); m.run(sensor, sps, runner=runner); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
