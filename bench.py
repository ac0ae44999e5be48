#!/usr/bin/env python
"""Headline benchmark: (snowpack x frequency) DORT solves per second on MI355X.

Workload (BASELINE.json configs[1], SURVEY.md 8d "cfg2"): IBA + DORT, 20 layers, 32 streams, 5 AMSR-E channels
(10.65/18.7/23.8/36.5/89 GHz, 55 deg, V+H), batch of 1024 synthetic snowpacks per GPU = 5120 solves per step.
A step = one pass of the hot path over that batch, inputs already resident in HBM (packed once, before timing).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

N > 1: one process per GPU, every rank solves its own 1024 x 5 batch (weak scaling, different seeds), the only
collective is the gather of the results to rank 0 (RCCL through torch.distributed), done inside every step.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_PEAK_TFLOPS = 78.6  # MI355X FP64 vector == matrix peak (AMD spec sheet); SURVEY.md 8(d)
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E ~8 TB/s
FLOPS_PER_N3 = 68.0  # algorithmic flops per layer = 68 N^3 (N = streams x polarisations)
FREQS = np.array([10.65e9, 18.7e9, 23.8e9, 36.5e9, 89e9])
N_SNOWPACKS = 1024
N_LAYERS = 20
N_STREAMS = 32
THETA_DEG = 55.0


def synthetic_snowpacks(seed, S=N_SNOWPACKS, L=N_LAYERS):
    """SURVEY.md 8(d) cfg2 laws: per snowpack draw thickness (L-1), density (L), temperature (L), size (L)."""
    rng = np.random.default_rng(seed)
    thick = np.empty((S, L))
    dens = np.empty((S, L))
    temp = np.empty((S, L))
    lc = np.empty((S, L))
    for s in range(S):
        thick[s, : L - 1] = rng.uniform(0.05, 0.30, L - 1)
        thick[s, L - 1] = 100.0
        dens[s] = rng.uniform(150, 450, L)
        temp[s] = rng.uniform(230, 270, L)
        lc[s] = rng.uniform(5e-5, 3e-4, L)
    return thick, dens, temp, lc


def _oracle_solve(args):
    from oracle import dort_oracle as O  # CPU baseline only

    thick, dens, temp, lc, f = args
    sp = dict(thickness=thick, density=dens, temperature=temp, microstructure="exponential", corr_length=lc)
    return O.solve(sp, f, [THETA_DEG], n_max_stream=N_STREAMS, method="half_rank_eig")


def cpu_baseline(thick, dens, temp, lc, gpu_values):
    """The CPU oracle (NumPy/SciPy restatement of the reference, its fastest diagonalisation method) on the host
    cores of this box, one worker process per core with BLAS threads = 1 (the reference's joblib default,
    smrt/runner/joblib_runner.py:18-31), on a bounded sample of the same workload."""
    import multiprocessing as mp

    for k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
        os.environ[k] = "1"
    cores = os.cpu_count() or 1
    cores = min(cores, 64)
    n_sample = int(min(512, max(64, 4 * cores)))
    S = thick.shape[0]
    items = []
    for i in range(n_sample):  # pairs in frequency-major order, like the GPU batch
        f, s = divmod(i * 37 % (len(FREQS) * S), S)
        items.append((thick[s], dens[s], temp[s], lc[s], FREQS[f], f, s))
    ctx = mp.get_context("fork")
    with ctx.Pool(cores) as pool:
        pool.map(_oracle_solve, [it[:5] for it in items[: 2 * cores]])  # warm the workers
        t0 = time.perf_counter()
        res = pool.map(_oracle_solve, [it[:5] for it in items], chunksize=1)
        dt = time.perf_counter() - t0
    err = 0.0
    for it, r in zip(items, res):
        err = max(err, float(np.abs(gpu_values[it[5] * S + it[6]] - r).max()))
    return dict(value=n_sample / dt, unit="solves/s", cores=cores, kind="port",
                sample="%d of the %d pairs of rank 0's batch, oracle/dort_oracle.py half_rank_eig, %d worker processes, "
                       "BLAS threads 1" % (n_sample, len(FREQS) * S, cores)), err


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--threads", type=int, default=0, help="workgroup size of the pair kernel (0 = library default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    torch = None
    # SMRT_BENCH_DIST=1: take the torch.distributed / RCCL path with a single rank too (tests/test_gpu_bench.py checks
    # the code the multi-GPU runs execute on a one-GPU box)
    use_dist = world > 1 or os.environ.get("SMRT_BENCH_DIST") == "1"
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        import torch
        import torch.distributed as dist

        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from smrt_amd._native import DortContext, PackedBatch

    thick, dens, temp, lc = synthetic_snowpacks(seed=2 + rank)
    batch = PackedBatch([N_LAYERS] * N_SNOWPACKS, thick, dens / 916.7, temp, lc, None, FREQS, np.deg2rad([THETA_DEG]),
                        emmodel="iba", microstructure="exponential", mode="P", n_max_stream=N_STREAMS)
    ctx = DortContext(local_rank)
    if args.threads:
        ctx.set_block_threads(args.threads)
    ctx.upload(batch)  # inputs resident in HBM before the timed region
    n_pairs = batch.n_pairs

    out_t = status_t = None
    gathered = [None, None]
    if use_dist:
        from smrt_amd.runner.distributed import gather_to_root

        out_t = torch.empty((n_pairs, 2), dtype=torch.float64, device="cuda")
        status_t = torch.empty((n_pairs,), dtype=torch.int32, device="cuda")

    def step():
        if use_dist:
            ctx.launch(out_t.data_ptr(), status_t.data_ptr())
            ctx.sync()  # the kernel runs on the context's own stream; RCCL runs on torch's
            gathered[0], gathered[1] = gather_to_root(dist, out_t, status_t, dst=0)  # the only collective
        else:
            ctx.launch()

    def fence():
        ctx.sync()
        if use_dist:
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    ctx.total_kernel_ms(reset=True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    kernel_ms_total, n_launch = ctx.total_kernel_ms()
    if use_dist:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    res = ctx.download()
    n_fail = int((res.status != 0).sum())
    if use_dist:
        res.values = out_t.cpu().numpy().reshape(res.values.shape)  # rank-local rows for the oracle cross-check
        if rank == 0:
            n_fail = int((gathered[1] != 0).sum().item())
    sum_n3 = ctx.sum_n3()
    flops_per_launch = FLOPS_PER_N3 * sum_n3
    kernel_ms = kernel_ms_total / max(n_launch, 1)
    achieved = flops_per_launch / (kernel_ms * 1e-3) / 1e12 if kernel_ms > 0 else 0.0

    traffic = None
    tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if os.path.exists(tpath):  # PMC counters cannot be collected from inside this process: measured with
        with open(tpath) as fh:  # tools/pmc_passes.sh on the same command, summary committed under profiles/
            tj = json.load(fh)
        if tj.get("solves_per_launch") == n_pairs:
            traffic = tj["traffic_bytes_per_launch"]

    if rank == 0:
        line = {
            "metric": "snowpack x frequency DORT solves/sec (20 layers, 32 streams)",
            "value": world * n_pairs * args.steps / elapsed,
            "unit": "solves/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": "BASELINE configs[1]: IBA + DORT passive, 20 layers, 32 streams, 5 AMSR-E channels "
                            "(10.65-89 GHz, 55 deg), 1024 synthetic snowpacks per GPU = 5120 solves per step",
                "solves_per_step_per_gpu": n_pairs,
                "parallelism": "%d independent rank(s), results gathered to rank 0" % world,
                "failed_solves": n_fail,
                "block_threads": 256,  # workgroup size of the three pipeline kernels (--threads only sizes the fused kernel)
            },
            "roofline": {
                "bound": "mfma",
                "achieved": achieved,
                "peak": FP64_PEAK_TFLOPS,
                "unit": "TFLOP/s",
                "frac": achieved / FP64_PEAK_TFLOPS,
                "traffic": traffic,
                "traffic_unit": "bytes per pipeline launch (rocprofv3 PMC FETCH_SIZE x2 + WRITE_SIZE, profiles/)",
                # the north star also asks for the fraction of the HBM roofline: measured bytes / pipeline time / 8 TB/s
                "hbm": (None if traffic is None or kernel_ms <= 0 else
                        {"achieved": traffic / (kernel_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": traffic / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS}),
                "kernel": "dort pipeline = dort_prep_kernel + dort_jacobi_kernel + dort_finish2_kernel (one launch each per "
                          "step; kernel_ms is their summed HIP-event time on the launch stream)",
                "kernel_ms": kernel_ms,
                "flops_per_launch": flops_per_launch,
                "note": "FP64 compute roofline (vector FMA rate = FP64 MFMA rate on gfx950, 78.6 TFLOP/s); algorithmic "
                        "flops = 68 * sum over pairs and layers of N_l^3 with the actual stream counts (SURVEY 8d); "
                        "algorithmic HBM bytes are ~1.6 KB per solve; the pipeline additionally stages ~75 KB per "
                        "(pair, layer) through HBM/L2 between its kernels and keeps F, G of the running layer there "
                        "(see DESIGN.md 4), still far from HBM-bound",
            },
        }
        if not args.no_cpu_baseline and world == 1:  # the CPU leg is reported at N = 1 only
            cb, err = cpu_baseline(thick, dens, temp, lc, res.values)
            line["cpu_baseline"] = cb
            line["config"]["max_abs_dTb_vs_oracle_K"] = err
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
