"""bench.py on the GPU box: the JSON contract of the single-GPU line, and the RCCL code path of the multi-GPU runs
(launched exactly like the driver launches it -- under torch.distributed.run, which is only the launcher: the ranks
rendezvous over a socket and gather with smrt_dort_gather -- with one rank, SMRT_BENCH_DIST=1)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _check_line(line, n_gpus):
    d = json.loads(line)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in d, key
    assert d["n_gpus"] == n_gpus and d["unit"] == "solves/s" and d["dtype"] == "f64" and d["scaling"] == "weak"
    assert d["config"]["failed_solves"] == 0 and "workload" in d["config"]
    r = d["roofline"]
    assert r["bound"] == "mfma" and 0.0 < r["frac"] < 1.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert abs(d["value"] - n_gpus * 5120 * d["steps"] / (d["ms_per_step"] * 1e-3 * d["steps"])) < 1e-6 * d["value"]
    return d


def test_bench_single_gpu_line():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = _check_line(lines[0], 1)
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0
    assert cb["blas_threads_per_worker"] == 1                      # measured inside the workers, not assumed
    assert 0 < cb["reference_default_method"]["value"] <= cb["value"] * 1.5
    assert d["config"]["max_abs_dTb_vs_oracle_K"] < 1e-6
    # the secondary rates of the same run: host buffers through smrt_dort_run, and the plugin surface end to end
    assert 0 < d["pcie_inclusive"]["value"] <= d["value"] * 1.05
    assert d["model_run"]["bitwise_equal_to_c_abi_run"] and 0 < d["model_run"]["value"] <= d["value"] * 1.05
    # per-kernel HIP-event times of three instrumented launches behind the timed region (smrt_dort_kernel_breakdown): the three
    # kinds add up to the pipeline's own event time within launch gaps, and carry SURVEY 8(d)'s flop shares
    pk = d["roofline"]["per_kernel"]
    assert set(pk) == {"prep", "diagonalise", "finish"} and all(v["ms"] > 0 for v in pk.values())
    # (measured one pipeline pass at a time; the timed region runs up to three concurrently and is a few percent faster)
    total = sum(v["ms"] for v in pk.values())
    assert 0.9 * d["roofline"]["kernel_ms"] < total <= 1.3 * d["roofline"]["kernel_ms"], (total, d["roofline"]["kernel_ms"])
    assert abs(sum(v["flop_share"] for v in pk.values()) - 1.0) < 1e-12
    # the kernels named are the ones smrt_dort_launch_info reports for this batch
    assert "dort_eig_tridiag_kernel" in d["roofline"]["kernel"] and "dort_finish_strip4_kernel" in d["roofline"]["kernel"]
    # the other two BASELINE shapes, short runs in the same command
    oc = d["other_configs"]
    assert set(oc) == {"2", "3"}
    for k, solves in (("2", 7168), ("3", 512)):
        assert oc[k]["solves_per_step"] == solves and oc[k]["failed_solves"] == 0 and oc[k]["value"] > 0
        assert 0.0 < oc[k]["roofline"]["frac"] < 1.0
    assert "dort_finish_strip_kernel" in oc["2"]["roofline"]["kernel"] and "big" in oc["3"]["roofline"]["kernel"]
    mr = d["model_run"]
    assert mr["first_run_ms"] >= 0.9 * mr["repeated_run_ms"] > 0


def test_bench_distributed_path_one_rank():
    """What `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` executes: socket rendezvous, RCCL
    communicator inside the library, the gather to rank 0 inside every step, max-over-ranks timing -- no torch import."""
    env = dict(os.environ, SMRT_BENCH_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3",
           "--warmup", "1", "--no-cpu-baseline"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = _check_line(lines[0], 1)
    assert "cpu_baseline" not in d


def test_bench_launches_its_own_ranks_or_refuses():
    """`python bench.py --gpus N` with no launcher in the environment starts the N ranks itself (RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* exported per rank, the socket rendezvous, RCCL inside the library) -- here with one rank
    (SMRT_BENCH_SPAWN=1 takes the launcher route for N = 1 too) -- and the line names the RCCL it ran on.  Asking for more
    ranks than there are GPUs, or a --gpus that contradicts the launcher's world, exits non-zero: a scaling run can never
    silently measure one rank."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    bench = os.path.join(ROOT, "bench.py")
    out = subprocess.run([sys.executable, bench, "--gpus", "1", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT,
                         env=dict(env, SMRT_BENCH_SPAWN="1", SMRT_BENCH_DIST="1"))
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = _check_line(lines[0], 1)
    assert "RCCL 2." in d["config"]["parallelism"] and "librccl" in d["config"]["parallelism"]
    # more ranks than GPUs (no node has 64; the library is not loaded into THIS process: a later test imports torch first)
    out = subprocess.run([sys.executable, bench, "--gpus", "64", "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert out.returncode != 0 and "visible" in out.stderr and not [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    # a launcher's world that is not --gpus
    out = subprocess.run([sys.executable, bench, "--gpus", "1", "--steps", "1", "--warmup", "0"], capture_output=True,
                         text=True, timeout=300, cwd=ROOT,
                         env=dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port())))
    assert out.returncode != 0 and "WORLD_SIZE=2" in out.stderr


def test_rccl_library_can_be_pinned():
    """smrt_dort_comm_library reports the RCCL the gather runs on; SMRT_RCCL_LIB pins it (a wrong pin fails loudly instead
    of falling back)."""
    code = ("import sys; sys.path.insert(0, %r); from smrt_amd._native import DortContext; "
            "print('|'.join(DortContext.comm_library()))" % ROOT)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    path, version = out.stdout.strip().split("|")
    assert os.path.exists(path) and "librccl" in path and version.startswith("2.")
    pinned = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300,
                            env=dict(os.environ, SMRT_RCCL_LIB=path))
    assert pinned.returncode == 0 and pinned.stdout.strip().split("|")[0] == path
    wrong = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300,
                           env=dict(os.environ, SMRT_RCCL_LIB="/nonexistent/librccl.so"))
    assert wrong.returncode != 0 and "SMRT_RCCL_LIB" in wrong.stderr


@pytest.mark.parametrize("config,metric", [(2, "50 layers, 64 streams"), (3, "active, 30 layers, 128 streams")])
def test_bench_other_configs_line(config, metric):
    """`bench.py --config 2 | 3`: the same line for the shapes of BASELINE configs[2] / configs[3] (small batches here)."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", str(config), "--snowpacks",
                          "32" if config == 2 else "8", "--steps", "1", "--warmup", "1"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert metric in d["metric"] and d["n_gpus"] == 1 and d["config"]["failed_solves"] == 0
    assert 0.0 < d["roofline"]["frac"] < 1.0 and d["roofline"]["traffic"] is None and "cpu_baseline" not in d


def test_launch_into_torch_buffers_matches_download():
    """smrt_dort_launch(out_dev, status_dev) with device pointers owned by torch (the multi-GPU bench does this)."""
    import torch

    from smrt_amd._native import DortContext, PackedBatch

    rng = np.random.default_rng(5)
    S, L = 6, 4
    b = PackedBatch([L] * S, rng.uniform(0.05, 0.3, (S, L)), rng.uniform(0.2, 0.45, (S, L)), rng.uniform(235, 268, (S, L)),
                    rng.uniform(5e-5, 3e-4, (S, L)), None, [18.7e9, 36.5e9], np.deg2rad([55.0]), n_max_stream=16)
    ctx = DortContext(0)
    try:
        ref = ctx.run(b)
        out_t = torch.full((b.n_pairs, 2), -1.0, dtype=torch.float64, device="cuda")
        st_t = torch.full((b.n_pairs,), -1, dtype=torch.int32, device="cuda")
        ctx.upload(b)
        ctx.launch(out_t.data_ptr(), st_t.data_ptr())
        ctx.sync()
        assert (st_t.cpu().numpy() == 0).all()
        assert np.array_equal(out_t.cpu().numpy().reshape(ref.values.shape), ref.values)
    finally:
        ctx.close()


def test_rccl_gather_single_process_communicator():
    """smrt_dort_comm_init_all (one process driving its GPUs, ncclCommInitAll) + smrt_dort_gather + allreduce_max on the
    one GPU of this box: the gathered rows are the downloaded rows, host and device side."""
    from smrt_amd._native import DortContext, PackedBatch

    rng = np.random.default_rng(6)
    S, L = 9, 5
    b = PackedBatch([L] * S, rng.uniform(0.05, 0.3, (S, L)), rng.uniform(0.2, 0.45, (S, L)), rng.uniform(235, 268, (S, L)),
                    rng.uniform(5e-5, 3e-4, (S, L)), None, [18.7e9, 36.5e9, 89e9], np.deg2rad([40.0, 55.0]), n_max_stream=16)
    ctx = DortContext(0)
    try:
        DortContext.comm_init_all([ctx])
        ctx.upload(b, 4, 20)
        cost = ctx.pair_cost()
        assert cost.shape == (20,) and (cost > 0).all()
        ctx.launch()
        v, s = ctx.gather([20], root=0)
        ref = ctx.download()
        assert np.array_equal(v, ref.values) and np.array_equal(s, ref.status) and (s == 0).all()
        # the work counter of the solve equals the estimate made before it (same stream counts)
        assert np.isclose(ctx.sum_n3(), cost.sum(), rtol=1e-12)
        assert list(ctx.allreduce_max([1.5, -2.0])) == [1.5, -2.0]
        ctx.barrier()
        with pytest.raises(Exception):
            ctx.gather([19], root=0)      # counts[own rank] must be the uploaded pair count
    finally:
        ctx.close()


@pytest.mark.gpu
def test_cfg3_shape_runs_on_the_three_kernel_pipeline():
    """50 layers x 64 streams (BASELINE configs[2]) must take the three-kernel pipeline on the global workspace, not the
    fused kernel (four times slower): the choice once hinged on an LDS plan of the FUSED kernel that misses the 160 KB
    by a few hundred bytes at 50 layers.  Asserted on what the library reports it runs (smrt_dort_launch_info), not on a
    wall-clock rate of a shared box."""
    from smrt_amd._native import DortContext, PackedBatch

    S, L = 128, 50
    rng = np.random.default_rng(3)
    thick = np.concatenate([rng.uniform(0.05, 0.3, (S, L - 1)), np.full((S, 1), 100.0)], axis=1)
    freqs = np.array([6.925e9, 7.3e9, 10.65e9, 18.7e9, 23.8e9, 36.5e9, 89e9])
    batch = PackedBatch([L] * S, thick, rng.uniform(150, 450, (S, L)) / 916.7, rng.uniform(230, 270, (S, L)),
                        rng.uniform(5e-5, 1.5e-4, (S, L)), np.full((S, L), 0.2), freqs, np.deg2rad([55.0]),
                        emmodel="dmrt_qca_shortrange", microstructure="sticky_hard_spheres", n_max_stream=64)
    ctx = DortContext(0)
    try:
        ctx.upload(batch)
        info = ctx.launch_info()
        # (passive, Flat interfaces: the strip finish kernel behind the prep and Jacobi kernels of the global-workspace pipeline)
        assert info["pipeline"] == "gmem_strip" and info["n_max"] == 128, info
        assert info["chunks"] * info["chunk_pairs"] >= batch.n_pairs > (info["chunks"] - 1) * info["chunk_pairs"], info
        ctx.launch(); ctx.sync()
        strip = ctx.download()
        assert (strip.status == 0).all()
        # ... and the pivoted finish kernel of that pipeline (set_pipeline(4)) gives the same brightness temperatures
        ctx.set_pipeline(4); ctx.upload(batch)
        assert ctx.launch_info()["pipeline"] == "gmem"
        ctx.launch(); ctx.sync()
        pivoted = ctx.download()
        ctx.set_pipeline(1)
        assert (pivoted.status == 0).all() and np.abs(strip.values - pivoted.values).max() < 1e-6
        # the other shapes take the kernels DESIGN.md section 4 names
        rng = np.random.default_rng(4)
        for n_stream, mode, want in ((32, "P", "lds_strip"), (16, "A", "lds_two_slot"), (96, "P", "big")):
            b = PackedBatch([4] * 8, rng.uniform(0.05, 0.3, (8, 4)), rng.uniform(0.2, 0.45, (8, 4)), rng.uniform(235, 268, (8, 4)),
                            rng.uniform(5e-5, 3e-4, (8, 4)), None, [18.7e9], np.deg2rad([40.0]), n_max_stream=n_stream, mode=mode)
            ctx.upload(b)
            assert ctx.launch_info()["pipeline"] == want, (n_stream, mode, ctx.launch_info())
        # deep snowpacks: the register-resident finish kernel's LDS grows with the layer count -- four of its wavefronts per
        # CU up to 40 layers at 32 streams, three beyond; it stays the default while three fit (160 KB / 3), and before that
        # limit is reached the per-layer tables of the OTHER kernels have already sent the batch to the global-workspace
        # pipeline: the kernel never runs at two per CU by default.  Same brightness temperatures as the
        # two-slot kernel at a depth where it runs three per CU.
        # The strip finish kernel (four wavefronts) is the default while three of its workgroups share a CU -- up to ~95 layers
        # at 32 streams --, then the register-resident one takes over under its own rule.
        lib = ctx._lib
        assert lib.smrt_dort_finish_reg_lds_bytes(32, 40) <= 40 * 1024 < lib.smrt_dort_finish_reg_lds_bytes(32, 41)
        assert lib.smrt_dort_finish_strip_lds_bytes(32, 60, 4) <= 160 * 1024 // 3 < lib.smrt_dort_finish_strip_lds_bytes(32, 110, 4)
        seen = {}
        for deep in (60, 93, 100, 150):
            th = rng.uniform(0.01, 0.05, (2, deep)); th[:, -1] = 100.0
            b = PackedBatch([deep] * 2, th, rng.uniform(0.2, 0.45, (2, deep)), rng.uniform(235, 268, (2, deep)),
                            rng.uniform(5e-5, 3e-4, (2, deep)), None, [18.7e9], np.deg2rad([40.0]), n_max_stream=32)
            ctx.upload(b)
            seen[deep] = ctx.launch_info()["pipeline"]
            if seen[deep] == "lds_reg":
                assert lib.smrt_dort_finish_reg_lds_bytes(32, deep) <= 160 * 1024 // 3, deep
            if deep == 93:   # the window in which a 32-stream batch runs on the eight-wavefront strip kernel (global-workspace pipeline)
                assert seen[deep] == "gmem_strip", seen
                ctx.launch(); ctx.sync(); strip8 = ctx.download()
                ctx.set_pipeline(4); ctx.upload(b)
                assert ctx.launch_info()["pipeline"] == "gmem"
                ctx.launch(); ctx.sync(); pivoted = ctx.download()
                ctx.set_pipeline(1)
                assert (strip8.status == 0).all() and np.abs(strip8.values - pivoted.values).max() < 1e-6
            if deep == 60:
                ctx.launch(); ctx.sync(); by_default = ctx.download()
                ctx.set_pipeline(4); ctx.upload(b)
                assert ctx.launch_info()["pipeline"] == "lds_two_slot"
                ctx.launch(); ctx.sync(); two_slot = ctx.download()
                ctx.set_pipeline(3); ctx.upload(b)
                assert ctx.launch_info()["pipeline"] == "lds_reg"
                ctx.launch(); ctx.sync(); reg = ctx.download()
                ctx.set_pipeline(1)
                assert (by_default.status == 0).all() and np.abs(by_default.values - two_slot.values).max() < 1e-6
                assert np.abs(by_default.values - reg.values).max() < 1e-6
        # (before the strip kernel's limit is reached the four LDS matrices of the FUSED plan have sent the batch to the
        # global-workspace pipeline -- 91 layers at 32 streams --, whose strip kernel needs the tables of 100 layers beside its
        # 136 KB matrix region: the pivoted kernel there)
        assert seen[60] == "lds_strip" and seen[100] in ("gmem", "gmem_strip") and seen[150] == "gmem", seen
    finally:
        ctx.close()


@pytest.mark.gpu
def test_kernel_breakdown_entry_point():
    """(Kept at the END of this module: the tests above that import torch must do so before the process has loaded the
    library -- two HIP runtimes in one process only get along in that order.)  smrt_dort_kernel_breakdown through ctypes: off by default (zeros), three positive intervals per chunk once enabled, on
    the strip pipeline of a small passive batch and on the fused path (nothing to report)."""
    from smrt_amd._native import DortContext, PackedBatch

    rng = np.random.default_rng(5)
    b = PackedBatch([4] * 64, rng.uniform(0.05, 0.3, (64, 4)), rng.uniform(0.2, 0.45, (64, 4)), rng.uniform(235, 268, (64, 4)),
                    rng.uniform(5e-5, 3e-4, (64, 4)), None, [18.7e9, 36.5e9], np.deg2rad([40.0]), n_max_stream=16)
    ctx = DortContext(0)
    try:
        ctx.upload(b); ctx.launch(); ctx.sync()
        off = ctx.kernel_breakdown()          # (reading leaves the instrumentation as it is: off)
        assert off["intervals"] == 0 and off["prep"] == off["jacobi"] == off["finish"] == 0.0
        ctx.launch()
        assert ctx.kernel_breakdown()["intervals"] == 0
        ctx.kernel_breakdown(True)
        ctx.launch()
        on = ctx.kernel_breakdown()
        ctx.sync()
        assert on["intervals"] >= 3 and min(on["prep"], on["jacobi"], on["finish"]) > 0.0
        assert on["prep"] + on["jacobi"] + on["finish"] <= 1.05 * ctx.last_kernel_ms() + 0.05
        ctx.kernel_breakdown(False)
        ctx.set_pipeline(0); ctx.upload(b); ctx.launch()      # fused kernel: no kinds to tell apart
        assert ctx.kernel_breakdown()["intervals"] == 0
        ctx.set_pipeline(1)
    finally:
        ctx.close()
