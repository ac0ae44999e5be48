#!/usr/bin/env python
"""Headline benchmark: (snowpack x frequency) DORT solves per second on MI355X.

Workload (BASELINE.json configs[1], SURVEY.md 8d "cfg2"): IBA + DORT, 20 layers, 32 streams, 5 AMSR-E channels
(10.65/18.7/23.8/36.5/89 GHz, 55 deg, V+H), batch of 1024 synthetic snowpacks per GPU = 5120 solves per step.
A step = one pass of the hot path over that batch with the inputs already resident in HBM (packed and uploaded once,
before timing); `value` is that rate.  The same JSON line also carries, measured in the same run:

* `pcie_inclusive`  -- the one-shot C entry point smrt_dort_run on host buffers: H2D of the packed inputs, the kernels,
                       D2H of results and diagnostics (SURVEY.md 8d counts the metric this way);
* `model_run`       -- Model.run -> HipBatchRunner -> Result on 1024 Snowpack objects: the plugin surface end to end;
* `roofline`        -- 68 sum N_l^3 flops / summed HIP-event time of the pipeline kernels, against the FP64 peak;
* `cpu_baseline`    -- the CPU oracle on this box's cores, one BLAS thread per worker (N = 1 only).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
        bench.py --gpus N --steps K --warmup W

N > 1: one process per GPU (only RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* are read from the launcher -- no PyTorch in
the path).  Started WITHOUT a launcher (`python bench.py --gpus N`, no WORLD_SIZE in the environment) it spawns the N ranks
itself; a world that does not match --gpus, or fewer visible GPUs than ranks, is an error (exit code 2), never a silent
one-rank run.  `--scaling weak` (default): every rank solves its own 1024 x 5 batch (different seeds); `--scaling strong`:
ONE 1024 x 5 batch cut into contiguous slices of equal estimated cost (sum N_l^3, smrt_dort_pair_cost).  Either way the
only collective is the gather of the result rows to rank 0 over RCCL (smrt_dort_gather), inside every step.
Prints ONE JSON line on rank 0.

`--config 2 | 3` runs the same measurement on the shapes of BASELINE configs[2] (DMRT-QCA-SR, 50 layers, 64 streams,
7 AMSR2 frequencies) and configs[3] (IBA active, Sentinel-1, 30 layers, 128 streams, m_max 2) with `--snowpacks S` per
GPU; the CPU baseline and the secondary measurements belong to the default configuration only.
"""
import os

# one BLAS / OpenMP thread per process, set BEFORE NumPy is imported anywhere in this process or its workers: the CPU
# baseline runs one worker per core like the reference's joblib runner (smrt/runner/joblib_runner.py:18-31,
# smrt/core/lib.py:655-666)
for _k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS", "NUMEXPR_NUM_THREADS"):
    os.environ[_k] = "1"

import argparse  # noqa: E402
import json  # noqa: E402
import sys  # noqa: E402
import time  # noqa: E402

import numpy as np  # noqa: E402

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


def synthetic_snowpacks(seed, S=N_SNOWPACKS, L=N_LAYERS, thick_range=(0.05, 0.30), last=100.0, size_range=(5e-5, 3e-4)):
    """SURVEY.md 8(d) laws: per snowpack draw thickness (L-1), density (L), temperature (L), size (L)."""
    rng = np.random.default_rng(seed)
    thick = np.empty((S, L))
    dens = np.empty((S, L))
    temp = np.empty((S, L))
    lc = np.empty((S, L))
    for s in range(S):
        thick[s, : L - 1] = rng.uniform(thick_range[0], thick_range[1], L - 1)
        thick[s, L - 1] = last
        dens[s] = rng.uniform(150, 450, L)
        temp[s] = rng.uniform(230, 270, L)
        lc[s] = rng.uniform(size_range[0], size_range[1], L)
    return thick, dens, temp, lc


def make_workload(config, rank, strong, n_snowpacks):
    """(PackedBatch, arrays, description) of BASELINE configs[config] (cfg2 / cfg3 / cfg4 there)."""
    from smrt_amd._native import PackedBatch

    if config == 1:
        S = n_snowpacks or N_SNOWPACKS
        arrays = synthetic_snowpacks(seed=2 if strong else 2 + rank, S=S)
        thick, dens, temp, lc = arrays
        batch = PackedBatch([N_LAYERS] * S, thick, dens / 916.7, temp, lc, None, FREQS, np.deg2rad([THETA_DEG]),
                            emmodel="iba", microstructure="exponential", mode="P", n_max_stream=N_STREAMS)
        return batch, arrays, dict(
            metric="snowpack x frequency DORT solves/sec (20 layers, 32 streams)",
            what="BASELINE configs[1]: IBA + DORT passive, 20 layers, 32 streams, 5 AMSR-E channels (10.65-89 GHz, 55 deg), "
                 "%d synthetic snowpacks" % S,
            )
    if config == 2:
        S, L = n_snowpacks or 1024, 50
        arrays = synthetic_snowpacks(seed=3 if strong else 3 + rank, S=S, L=L, size_range=(5e-5, 1.5e-4))
        thick, dens, temp, radius = arrays
        freqs = np.array([6.925e9, 7.3e9, 10.65e9, 18.7e9, 23.8e9, 36.5e9, 89e9])
        batch = PackedBatch([L] * S, thick, dens / 916.7, temp, radius, np.full((S, L), 0.2), freqs, np.deg2rad([55.0]),
                            emmodel="dmrt_qca_shortrange", microstructure="sticky_hard_spheres", n_max_stream=64)
        return batch, arrays, dict(
            metric="snowpack x frequency DORT solves/sec (50 layers, 64 streams)",
            what="BASELINE configs[2] shape: DMRT-QCA short-range + DORT passive, 50 layers, 64 streams, 7 AMSR2 frequencies "
                 "(stickiness 0.2), %d synthetic snowpacks" % S,
            )
    if config == 3:
        S, L = n_snowpacks or 512, 30
        arrays = synthetic_snowpacks(seed=4 if strong else 4 + rank, S=S, L=L, thick_range=(0.02, 0.10), last=1000.0)
        thick, dens, temp, lc = arrays
        batch = PackedBatch([L] * S, thick, dens / 916.7, temp, lc, None, [5.405e9], np.deg2rad(np.arange(20.0, 46.0, 5.0)),
                            emmodel="iba", microstructure="exponential", mode="A", n_max_stream=128, m_max=2)
        return batch, arrays, dict(
            metric="snowpack x frequency DORT solves/sec (active, 30 layers, 128 streams, m_max 2)",
            what="BASELINE configs[3] shape: IBA + DORT active, Sentinel-1 C band (5.405 GHz, 20..45 deg), 30 layers, "
                 "128 streams, m_max 2, %d synthetic snowpacks" % S,
            )
    raise SystemExit("bench.py: --config must be 1, 2 or 3")


PIPELINE_KERNELS = {   # smrt_dort_launch_info's pipeline -> the kernels of a launch (prep, finish; the diagonalisation between them below)
    "fused": ("dort_fused_kernel (everything of a pair in one workgroup)", None),
    "lds_two_slot": ("dort_prep_kernel", "dort_finish2_kernel"),
    "lds_four_slot": ("dort_prep_kernel", "dort_finish_kernel"),
    "lds_reg": ("dort_prep_kernel", "dort_finish_reg_kernel"),
    "lds_strip": ("dort_prep_kernel", "dort_finish_strip4_kernel"),
    "fused_gmem": ("dort_fused_gmem_kernel (everything of a pair in one workgroup, global workspace)", None),
    "gmem": ("dort_prep_kernel_gmem", "dort_finish_kernel_gmem"),
    "gmem_strip": ("dort_prep_kernel_wide", "dort_finish_strip_kernel"),
    "big": ("dort_passive_big_kernel / dort_active_big_kernel (prep)", "dort_passive_big_kernel / dort_active_big_kernel (finish)"),
}
RAYLEIGH_KERNEL = ("dort_rayleigh_kernel (layers with a Rayleigh phase matrix: diagonal-minus-rank-two eigenproblem in closed form -- "
                   "no Cholesky, no iteration)")
DIAG_KERNELS = {
    "jacobi": "dort_jacobi_kernel (one launch per size class of items; dort_jacobi_big_kernel above 128 rows)",
    "symmetric": "dort_eig_gram_kernel + dort_eig_tridiag_kernel + dort_eig_chase_kernel + dort_eig_vectors_kernel "
                 "(symmetric eigensolver of B B^T, one launch per size class)",
}


def describe_kernels(info):
    """What a launch of the uploaded batch consists of, from smrt_dort_launch_info (not a hard-coded string)."""
    prep, finish = PIPELINE_KERNELS[info["pipeline"]]
    if finish is None:
        return "pipeline '%s': %s" % (info["pipeline"], prep)
    if info["pipeline"] == "lds_strip" and info.get("rayleigh_closed_form"):
        finish = "dort_finish_strip4_direct_kernel"   # (the instance that also takes layers diagonalised in closed form)
    diag = RAYLEIGH_KERNEL if info.get("rayleigh_closed_form") else DIAG_KERNELS[info["diagonalisation"]]
    return "pipeline '%s': %s + %s + %s; %d pipeline pass(es) of <= %d pairs per launch" % (
        info["pipeline"], prep, diag, finish, info["chunks"], info["chunk_pairs"])


def algorithmic_flops(sum_n3, info, all_rayleigh):
    """(flops per launch, note).  SURVEY 8(d) books 68 N^3 per layer on the reference's algorithm: ~2 (assembly, Cholesky,
    B) + ~25 (diagonalisation) + ~41 (eigenvector recovery, layer recursion).  Where every layer of the batch has a
    Rayleigh phase matrix and the closed-form kernel diagonalises it, the first two parts are O(N^2) work that is not
    counted at all: only the 41 N^3 of the recursion are -- the roofline fraction prices what the device executes, not
    flops it was spared (with all 68 it would read above 1)."""
    if info.get("rayleigh_closed_form") and all_rayleigh:
        return 41.0 * sum_n3, "41 N^3 per layer (closed-form diagonalisation of the Rayleigh layers: the reference's 2 + 25 N^3 of assembly / Cholesky / diagonalisation are O(N^2) here and not counted)"
    return FLOPS_PER_N3 * sum_n3, "68 N^3 per layer (SURVEY 8d)"


# ---- CPU baseline ------------------------------------------------------------------------------------------------
def _worker_init():
    try:  # belt and braces on top of the environment variables: cap whatever BLAS this worker loaded
        from threadpoolctl import threadpool_limits

        threadpool_limits(1)
    except Exception:  # noqa: BLE001
        pass


def _worker_threads(_):
    try:
        from threadpoolctl import threadpool_info

        info = threadpool_info()
        return max([p.get("num_threads", 1) for p in info] or [1])
    except Exception:  # noqa: BLE001
        return -1


def _oracle_solve(args):
    from oracle import dort_oracle as O  # CPU baseline only

    thick, dens, temp, lc, f, method = args
    sp = dict(thickness=thick, density=dens, temperature=temp, microstructure="exponential", corr_length=lc)
    return O.solve(sp, f, [THETA_DEG], n_max_stream=N_STREAMS, method=method)


def cpu_baseline(thick, dens, temp, lc, gpu_values):
    """The CPU oracle (NumPy/SciPy restatement of the reference) on the host cores of this box: one worker process per
    core, one BLAS thread each (measured in the workers and reported), on a bounded sample of the same workload; with
    the oracle's fastest diagonalisation (`half_rank_eig`, the headline number) and with the reference's default
    (`schur_forcedtriu`)."""
    import multiprocessing as mp

    cores = min(os.cpu_count() or 1, 64)
    S = thick.shape[0]

    def items(n, method):
        out = []
        for i in range(n):  # pairs spread over the frequency-major list of the GPU batch
            f, s = divmod(i * 37 % (len(FREQS) * S), S)
            out.append(((thick[s], dens[s], temp[s], lc[s], FREQS[f], method), f, s))
        return out

    n_fast = int(min(512, max(64, 4 * cores)))
    n_ref = int(min(256, max(32, 2 * cores)))
    rates = {}
    err = 0.0
    with mp.get_context("spawn").Pool(cores, initializer=_worker_init) as pool:
        threads = max(pool.map(_worker_threads, range(cores)))
        pool.map(_oracle_solve, [it[0] for it in items(cores, "half_rank_eig")])  # warm the workers (imports)
        for method, n in (("half_rank_eig", n_fast), ("schur_forcedtriu", n_ref)):
            its = items(n, method)
            t0 = time.perf_counter()
            res = pool.map(_oracle_solve, [it[0] for it in its], chunksize=1)
            rates[method] = n / (time.perf_counter() - t0)
            for it, r in zip(its, res):
                err = max(err, float(np.abs(gpu_values[it[1] * S + it[2]] - r).max()))
    return dict(value=rates["half_rank_eig"], unit="solves/s", cores=cores, kind="port",
                blas_threads_per_worker=threads,
                reference_default_method=dict(value=rates["schur_forcedtriu"], unit="solves/s", method="schur_forcedtriu",
                                              sample=n_ref),
                sample="%d of the %d pairs of rank 0's batch, oracle/dort_oracle.py half_rank_eig, %d spawned worker "
                       "processes, %d BLAS thread(s) per worker (threadpoolctl)" % (n_fast, len(FREQS) * S, cores, threads)), err


# ---- secondary measurements on rank 0 ----------------------------------------------------------------------------
def pcie_inclusive(ctx, batch, reps=5):
    ctx.run(batch)
    times = []
    for _ in range(reps):
        t0 = time.perf_counter()
        ctx.run(batch)  # smrt_dort_run: H2D + kernels + D2H (values, status, layer and stream diagnostics)
        times.append(time.perf_counter() - t0)
    ms = 1e3 * float(np.median(times))
    return dict(value=batch.n_pairs / (ms * 1e-3), unit="solves/s", ms=ms,
                what="smrt_dort_run on host buffers: H2D of the packed inputs, kernels, D2H of results + diagnostics; "
                     "median of %d" % reps)


def model_run_rate(thick, dens, temp, lc, reference_values, reps=3):
    from smrt_amd import make_model, make_snowpack
    from smrt_amd.core.sensor import passive

    t0 = time.perf_counter()
    sps = [make_snowpack(thick[s], "exponential", density=dens[s], temperature=temp[s], corr_length=lc[s])
           for s in range(thick.shape[0])]
    build_s = time.perf_counter() - t0
    sensor = passive(list(FREQS), THETA_DEG)
    m = make_model("iba", "dort", rtsolver_options=dict(n_max_stream=N_STREAMS, devices=[int(os.environ.get("LOCAL_RANK", "0"))]))
    t0 = time.perf_counter()
    res = m.run(sensor, sps)
    first_ms = 1e3 * (time.perf_counter() - t0)
    times = []
    for _ in range(reps):
        t0 = time.perf_counter()
        res = m.run(sensor, sps)
        times.append(time.perf_counter() - t0)
    ms = 1e3 * float(np.median(times))
    same = bool(np.array_equal(res.data.values.reshape(reference_values.shape), reference_values))
    return dict(value=len(sps) * len(FREQS) / (ms * 1e-3), unit="solves/s", ms=ms, repeated_run_ms=ms, first_run_ms=first_ms,
                first_run_value=len(sps) * len(FREQS) / (first_ms * 1e-3), bitwise_equal_to_c_abi_run=same,
                snowpack_objects_built_in_s=build_s,
                what="make_model('iba','dort').run(passive(5 freqs), 1024 Snowpack objects) -> stacked Result: `first_run_ms` is "
                     "the first call on fresh objects (context creation, packing and the device's buffer allocation included), "
                     "`value` / `repeated_run_ms` the median of %d more calls on the same objects (built once, outside)" % reps)


def other_config(config, n_snowpacks, steps, local_rank):
    """A short driver-timed run of another BASELINE shape on its own context (inputs resident, like the headline)."""
    from smrt_amd._native import DortContext

    batch, _, desc = make_workload(config, 0, False, n_snowpacks)
    ctx = DortContext(local_rank)
    try:
        ctx.upload(batch)
        ctx.launch(); ctx.sync()      # warm-up (first-touch of the staging area)
        ctx.total_kernel_ms(reset=True)
        t0 = time.perf_counter()
        for _ in range(steps):
            ctx.launch()
        ctx.sync()
        elapsed = time.perf_counter() - t0
        kernel_ms_total, n_launch = ctx.total_kernel_ms()
        res = ctx.download()
        kernel_ms = kernel_ms_total / max(n_launch, 1)
        info = ctx.launch_info()
        flops, flops_note = algorithmic_flops(ctx.sum_n3(), info, config == 2)
        return dict(workload=desc["what"], value=batch.n_pairs * steps / elapsed, unit="solves/s", steps=steps, warmup=1,
                    ms_per_step=1e3 * elapsed / steps, solves_per_step=batch.n_pairs, failed_solves=int((res.status != 0).sum()),
                    roofline=dict(bound="mfma", achieved=flops / (kernel_ms * 1e-3) / 1e12, peak=FP64_PEAK_TFLOPS, unit="TFLOP/s",
                                  frac=flops / (kernel_ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS, kernel_ms=kernel_ms,
                                  flops_per_launch=flops, flops_counted=flops_note, kernel=describe_kernels(info)))
    finally:
        ctx.close()


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves -- one process per GPU with the
    variables a launcher would export -- pass rank 0's line through and return the worst exit code."""
    import socket
    import subprocess

    from smrt_amd._native import device_count

    visible = device_count()
    if visible < n:
        sys.stderr.write("bench.py: --gpus %d but only %d GPU(s) are visible: refusing to measure fewer ranks than asked\n" % (n, visible))
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    worst = 0
    try:
        for p in procs:
            worst = max(worst, abs(p.wait()))
            if worst:
                break
    finally:
        for p in procs:      # (exactly the processes started above)
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p.kill()
    return worst


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default 20; 5 / 3 for --config 2 / 3)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed steps (default 3; 1 for --config 2 / 3)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak")
    ap.add_argument("--config", type=int, default=1, help="BASELINE configs[1] (default, the headline) | [2] | [3]")
    ap.add_argument("--snowpacks", type=int, default=0, help="snowpacks per GPU (default 1024 / 1024 / 512 by --config)")
    ap.add_argument("--threads", type=int, default=0, help="workgroup size of the per-pair kernels (0 = library default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the pcie_inclusive / model_run / other_configs measurements")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the short runs of configs[2] and configs[3] on the headline line")
    args = ap.parse_args()
    if args.steps is None:
        args.steps = {1: 20, 2: 5, 3: 3}.get(args.config, 20)
    if args.warmup is None:
        args.warmup = 3 if args.config == 1 else 1
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be at least 1")

    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or os.environ.get("SMRT_BENCH_SPAWN") == "1"):
        # no launcher: be the launcher (SMRT_BENCH_SPAWN=1 takes this route with one rank too -- tests/test_gpu_bench.py)
        sys.exit(spawn_ranks(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks\n" % (args.gpus, world))
        sys.exit(2)
    # SMRT_BENCH_DIST=1: take the RCCL path with a single rank too (tests/test_gpu_bench.py checks on a one-GPU box the
    # code the multi-GPU runs execute)
    use_comm = world > 1 or os.environ.get("SMRT_BENCH_DIST") == "1"
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    from smrt_amd._native import DortContext, device_count
    from smrt_amd.rtsolver.dort import shard_by_cost
    from smrt_amd.runner.distributed import init_comm

    if device_count() <= local_rank:
        sys.stderr.write("bench.py: rank %d needs GPU %d but %d GPU(s) are visible\n" % (rank, local_rank, device_count()))
        sys.exit(2)
    strong = args.scaling == "strong"
    batch, (thick, dens, temp, lc), desc = make_workload(args.config, rank, strong, args.snowpacks)
    headline = args.config == 1 and not args.snowpacks
    ctx = DortContext(local_rank)
    if args.threads:
        ctx.set_block_threads(args.threads)
    rccl = None
    if use_comm:
        init_comm(ctx, rank, world)
        rccl = DortContext.comm_library()     # (path the symbols were resolved from, version); SMRT_RCCL_LIB pins it
    lo, hi = 0, batch.n_pairs
    if strong and world > 1:  # equal estimated cost per rank, contiguous slices of the frequency-major list
        ctx.upload(batch)
        bounds = shard_by_cost(ctx.pair_cost(), world)
        counts = np.diff(bounds)
        if (counts <= 0).any():   # every rank decides the same way (same costs): nobody is left waiting in the gather
            raise SystemExit("bench.py --scaling strong: %d ranks for %d pairs leaves a rank without work" % (world, batch.n_pairs))
        lo, hi = int(bounds[rank]), int(bounds[rank + 1])
    else:
        counts = np.full(world, batch.n_pairs, np.int64)
    ctx.upload(batch, lo, hi - lo)  # inputs resident in HBM before the timed region
    n_pairs = hi - lo
    total_pairs = int(counts.sum())
    gathered = [None, None]

    def step():
        ctx.launch()
        if use_comm:
            gathered[0], gathered[1] = ctx.gather(counts, root=0)  # the only collective; waits for the kernels

    def fence():
        ctx.sync()
        if use_comm:
            ctx.barrier()
            ctx.sync()

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
    per_rank = None
    if use_comm:
        elapsed = float(ctx.allreduce_max([elapsed])[0])   # MAX over ranks
        # every rank's kernel time and pair count in rank 0's line (a vector with one own entry each, MAX-reduced):
        # load imbalance shows in the first scaling record
        mine = np.zeros(2 * world)
        mine[rank] = kernel_ms_total / max(n_launch, 1)
        mine[world + rank] = n_pairs
        allv = ctx.allreduce_max(mine)
        per_rank = {"kernel_ms": [float(v) for v in allv[:world]], "pairs": [int(v) for v in allv[world:]]}

    # which kernel the time goes to: three more launches with an event pair around every kernel (outside the timed region)
    res = ctx.download()
    info = ctx.launch_info()
    per_kernel = None
    if not use_comm:
        # (one pipeline pass at a time for this: the intervals of concurrent passes overlap and would not add up)
        lanes_env = os.environ.get("SMRT_DORT_LANES")
        os.environ["SMRT_DORT_LANES"] = "1"
        ctx.upload(batch, lo, hi - lo)
        ctx.launch(); ctx.sync()
        ctx.kernel_breakdown(True)
        acc = {"prep": 0.0, "jacobi": 0.0, "finish": 0.0}
        for _ in range(3):
            ctx.launch()
            bd = ctx.kernel_breakdown()
            for k in acc:
                acc[k] += bd[k] / 3.0
        ctx.kernel_breakdown(False)
        if lanes_env is None:
            del os.environ["SMRT_DORT_LANES"]
        else:
            os.environ["SMRT_DORT_LANES"] = lanes_env
        ctx.upload(batch, lo, hi - lo)
        if sum(acc.values()) > 0:
            per_kernel = {"prep": {"ms": acc["prep"]}, "diagonalise": {"ms": acc["jacobi"]}, "finish": {"ms": acc["finish"]}}
    n_fail = int((res.status != 0).sum())
    if use_comm and rank == 0:
        n_fail = int((gathered[1] != 0).sum())
        own = gathered[0][: n_pairs]  # rank 0's rows come first
        assert np.array_equal(own, res.values), "gathered rows differ from the local ones"
    sum_n3 = ctx.sum_n3()
    flops_per_launch, flops_note = algorithmic_flops(sum_n3, info, args.config == 2)
    kernel_ms = kernel_ms_total / max(n_launch, 1)
    achieved = flops_per_launch / (kernel_ms * 1e-3) / 1e12 if kernel_ms > 0 else 0.0

    traffic = None
    tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if os.path.exists(tpath):  # PMC counters cannot be collected from inside this process: measured with
        with open(tpath) as fh:  # tools/pmc_passes.sh on the same command, summary committed under profiles/
            tj = json.load(fh)
        if args.config != 1:        # the other shapes: their own PMC passes (PMC_CONFIG=2|3 tools/pmc_passes.sh)
            tj = tj.get("configs", {}).get(str(args.config), {})
        if tj.get("solves_per_launch") == n_pairs and (headline or args.config != 1):
            traffic = tj["traffic_bytes_per_launch"]

    if rank == 0:
        line = {
            "metric": desc["metric"],
            "value": total_pairs * args.steps / elapsed,
            "unit": "solves/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": "%s %s = %d solves per step"
                            % (desc["what"], "in all, cut by estimated cost over the ranks" if strong else "per GPU", total_pairs),
                "solves_per_step_per_gpu": n_pairs,
                "parallelism": "%d independent rank(s), results gathered to rank 0%s" % (
                    world, " by smrt_dort_gather (RCCL %s from %s, no PyTorch)" % (rccl[1], rccl[0]) if use_comm else ""),
                "failed_solves": n_fail,
                "timed_region": "smrt_dort_launch of the resident batch (+ smrt_dort_gather when N > 1), "
                                "barrier + stream sync on both sides, max over ranks",
                "value_is": "the resident-input rate (inputs in HBM when the timed region starts, as the bench contract asks); "
                            "SURVEY 8(d)'s rate including H2D of the packed inputs and D2H of the results is `pcie_inclusive` "
                            "of this line, the plugin surface end to end is `model_run`",
            },
            "roofline": {
                "bound": "mfma",
                "achieved": achieved,
                "peak": FP64_PEAK_TFLOPS,
                "unit": "TFLOP/s",
                "frac": achieved / FP64_PEAK_TFLOPS,
                "traffic": traffic,
                "traffic_unit": "bytes per pipeline launch (rocprofv3 PMC FETCH_SIZE x2 + WRITE_SIZE, profiles/)",
                "traffic_source": ("profiles/hbm_traffic.json (round %s; separate rocprofv3 --pmc passes over this command, "
                                   "tools/pmc_passes.sh -- not measured in this run)" % tj.get("round")) if traffic is not None else None,
                # the north star also asks for the fraction of the HBM roofline: measured bytes / pipeline time / 8 TB/s
                "hbm": (None if traffic is None or kernel_ms <= 0 else
                        {"achieved": traffic / (kernel_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": traffic / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS}),
                "kernel": describe_kernels(info) + " (kernel_ms: HIP events around the whole launch on the context's stream)",
                "per_rank": per_rank,
                # HIP-event time per kernel kind (three instrumented launches after the timed region, ONE pipeline pass at a
                # time -- the timed region may run several concurrently, so these need not add up to ms_per_step) with the
                # share of the algorithmic flops SURVEY 8(d) books on it: assembly + Cholesky x 2 + B = L+^T L- ~ 2 N^3
                # (prep), the diagonalisation ~ 25 N^3 (the reference's count, whatever algorithm runs), eigenvector
                # recovery + layer recursion ~ 41 N^3 (finish)
                "per_kernel": (None if per_kernel is None else
                               {k: dict(v, flop_share=fs, tflops=fs * flops_per_launch / (v["ms"] * 1e-3) / 1e12 if v["ms"] > 0 else None)
                                for (k, v), fs in zip(per_kernel.items(), (0.0, 0.0, 1.0) if flops_note.startswith("41") else
                                                      (2.0 / 68.0, 25.0 / 68.0, 41.0 / 68.0))}),
                "kernel_ms": kernel_ms,
                "flops_per_launch": flops_per_launch,
                "flops_counted": flops_note,
                "note": "FP64 compute roofline: vector FMA and FP64 MFMA share one 78.6 TFLOP/s pipe on gfx950 "
                        "(tools/micro/fp64_pipes.hip: 60 / 71 / 72 TFLOP/s alone / alone / together); algorithmic flops = "
                        "68 * sum over pairs and layers of N_l^3 with the actual stream counts (SURVEY 8d); algorithmic "
                        "HBM bytes are ~1.6 KB per solve; the pipeline additionally stages ~75 KB per (pair, layer) "
                        "through HBM/L2 between its kernels (DESIGN.md 4), still far from HBM-bound",
            },
        }
        if world == 1 and headline and not args.no_secondary:
            line["pcie_inclusive"] = pcie_inclusive(ctx, batch)
            line["model_run"] = model_run_rate(thick, dens, temp, lc, res.values)
        if not args.no_cpu_baseline and world == 1 and headline:  # the CPU leg is reported at N = 1 only
            cb, err = cpu_baseline(thick, dens, temp, lc, res.values)
            line["cpu_baseline"] = cb
            line["config"]["max_abs_dTb_vs_oracle_K"] = err
        assert line["n_gpus"] == args.gpus
        if world == 1 and not use_comm and headline and not args.no_secondary and not args.no_other_configs:
            # the other two BASELINE shapes, timed by whoever runs this command (short: 2 steps of configs[2] at 1024
            # snowpacks, 1 step of configs[3] at 512 -- their own contexts, this one's buffers released first)
            ctx.close()
            line["other_configs"] = {"2": other_config(2, 1024, 2, local_rank), "3": other_config(3, 512, 1, local_rank)}
        print(json.dumps(line), flush=True)
    if use_comm:
        ctx.barrier()
    ctx.close()


if __name__ == "__main__":
    main()
