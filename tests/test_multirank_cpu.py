"""World-size-2 test of the multi-rank path on CPU (gloo): every rank solves its contiguous shard of one batch --
with the device code running under the CPU emulator, since there is no GPU here -- and the results are gathered to
rank 0 with the same helper bench.py / a multi-process runner use with RCCL.  The gathered result must equal the
single-process result bit for bit."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

from conftest import ROOT

EMU_LIB = os.path.join(ROOT, "tests", "hostemu", "libsmrt_emu.so")


def _make_batch():
    from smrt_amd._native import PackedBatch

    rng = np.random.default_rng(21)
    S, L = 3, 4
    thick = np.concatenate([rng.uniform(0.05, 0.3, (S, L - 1)), np.full((S, 1), 100.0)], axis=1)
    return PackedBatch([L] * S, thick, rng.uniform(150, 450, (S, L)) / 916.7, rng.uniform(230, 270, (S, L)),
                       rng.uniform(5e-5, 3e-4, (S, L)), None, [18.7e9, 36.5e9, 89e9], np.deg2rad([55.0]), n_max_stream=8)


def _emu_run(batch, lo, n):
    from smrt_amd._native import SmrtBatch

    lib = C.CDLL(EMU_LIB)
    P = C.POINTER
    lib.smrt_emu_run.argtypes = [P(SmrtBatch), C.c_longlong, C.c_longlong, C.c_int, C.c_int, P(C.c_double),
                                 P(C.c_int32), P(C.c_double), P(C.c_double), P(C.c_double), P(C.c_long)]
    out = np.empty((n,) + batch.out_shape())
    st = np.empty(n, np.int32)
    rc = lib.smrt_emu_run(C.byref(batch.struct), lo, n, 64, 0, out.ctypes.data_as(P(C.c_double)),
                          st.ctypes.data_as(P(C.c_int32)), None, None, None, None)
    assert rc == 0
    return out, st


def _worker(rank, world, port, tmpdir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist

    from smrt_amd.runner.distributed import gather_to_root, shard_bounds

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    batch = _make_batch()
    b = shard_bounds(batch.n_pairs, world)
    out, st = _emu_run(batch, int(b[rank]), int(b[rank + 1] - b[rank]))
    v, s = gather_to_root(dist, torch.from_numpy(out), torch.from_numpy(st), dst=0)
    dist.barrier()
    if rank == 0:
        full, fst = _emu_run(batch, 0, batch.n_pairs)
        ok = np.array_equal(v.numpy(), full) and np.array_equal(s.numpy(), fst) and (fst == 0).all()
        open(os.path.join(tmpdir, "result.txt"), "w").write("OK" if ok else "MISMATCH")
    dist.destroy_process_group()


def test_shard_bounds():
    from smrt_amd.runner.distributed import shard_bounds

    b = shard_bounds(5120, 8)
    assert b[0] == 0 and b[-1] == 5120 and (np.diff(b) == 640).all()
    b = shard_bounds(9, 2)
    assert list(b) == [0, 4, 9]
    assert list(shard_bounds(3, 8)) == [0, 0, 0, 1, 1, 1, 2, 2, 3] or shard_bounds(3, 8)[-1] == 3


def test_two_ranks_gloo_gather(tmp_path):
    if not os.path.exists(EMU_LIB):
        pytest.skip("emulator library not built (python __graft_entry__.py)")
    import torch.multiprocessing as mp

    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert open(os.path.join(tmp_path, "result.txt")).read() == "OK"


def _rdzv_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world), RANK=str(rank))
    from smrt_amd.runner.distributed import broadcast_from_root

    payload = bytes(range(128)) if rank == 0 else None
    q.put((rank, broadcast_from_root(payload, rank, world, timeout=60.0)))


def test_socket_rendezvous_hands_the_id_to_every_rank():
    """The PyTorch-free rendezvous of the multi-GPU path (smrt_amd/runner/distributed.py): rank 0 serves 128 bytes (the
    RCCL unique id in production) on a port next to MASTER_PORT, three other processes fetch it -- late starters and a
    stranger on the port range included."""
    import multiprocessing as mp
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    squatter = socket.socket()          # somebody else already listens on the first candidate port
    try:
        squatter.bind(("127.0.0.1", port + 1))
        squatter.listen(1)
    except OSError:
        pass
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    world = 4
    procs = [ctx.Process(target=_rdzv_worker, args=(r, world, port, q)) for r in (2, 1, 3)]
    for p in procs:
        p.start()
    root = ctx.Process(target=_rdzv_worker, args=(0, world, port, q))   # the root starts last
    root.start()
    got = dict(q.get(timeout=90) for _ in range(world))
    for p in procs + [root]:
        p.join(30)
    squatter.close()
    assert set(got) == {0, 1, 2, 3} and all(v == bytes(range(128)) for v in got.values())
