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


def _emu_run(batch, lo, n, want_cost=False):
    from smrt_amd._native import SmrtBatch

    lib = C.CDLL(EMU_LIB)
    P = C.POINTER
    lib.smrt_emu_run.argtypes = [P(SmrtBatch), C.c_longlong, C.c_longlong, C.c_int, C.c_int, P(C.c_double),
                                 P(C.c_int32), P(C.c_double), P(C.c_double), P(C.c_double), P(C.c_long)]
    out = np.empty((n,) + batch.out_shape())
    st = np.empty(n, np.int32)
    n3 = np.zeros(n)
    rc = lib.smrt_emu_run(C.byref(batch.struct), lo, n, 64, 0, out.ctypes.data_as(P(C.c_double)),
                          st.ctypes.data_as(P(C.c_int32)), None, None, n3.ctypes.data_as(P(C.c_double)) if want_cost else None,
                          None)
    assert rc == 0
    return (out, st, n3) if want_cost else (out, st)


def _worker(rank, world, port, tmpdir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist

    from smrt_amd._native import gather_plan
    from smrt_amd.rtsolver.dort import shard_by_cost

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    batch = _make_batch()
    # what the product shards by (smrt_dort_pair_cost: sum over the layers of N_l^3), here from the emulated solve of the
    # whole batch; the slices are unequal in length
    full, fst, cost = _emu_run(batch, 0, batch.n_pairs, want_cost=True)
    b = shard_by_cost(cost, world)
    counts = np.diff(b)
    out, st = _emu_run(batch, int(b[rank]), int(counts[rank]))
    # the product's collective is grouped send / recv following smrt_dort_gather_plan (dort_comm.hip): the SAME plan drives
    # gloo send / recv here, with the root that is not rank 0 so that the own-offset arithmetic matters
    root = world - 1
    ops, own_off, total = gather_plan(world, root, rank, counts)
    stride = int(np.prod(batch.out_shape()))
    if rank != root:
        for peer, _, rows in ops:
            dist.send(torch.from_numpy(out.reshape(-1, stride)[:rows].copy()), dst=peer)
            dist.send(torch.from_numpy(st[:rows].copy()), dst=peer)
    else:
        gv = torch.zeros((total, stride), dtype=torch.float64)
        gs = torch.full((total,), -1, dtype=torch.int32)
        for peer, off, rows in ops:
            dist.recv(gv[off:off + rows], src=peer)
            dist.recv(gs[off:off + rows], src=peer)
        gv[own_off:own_off + counts[root]] = torch.from_numpy(out.reshape(-1, stride))
        gs[own_off:own_off + counts[root]] = torch.from_numpy(st)
        ok = (np.array_equal(gv.numpy().reshape(full.shape), full) and np.array_equal(gs.numpy(), fst) and (fst == 0).all()
              and len(set(counts.tolist())) > 1)
        open(os.path.join(tmpdir, "result.txt"), "w").write("OK" if ok else "MISMATCH %s" % counts)
    dist.barrier()
    dist.destroy_process_group()


def test_shard_bounds():
    from smrt_amd.runner.distributed import shard_bounds

    b = shard_bounds(5120, 8)
    assert b[0] == 0 and b[-1] == 5120 and (np.diff(b) == 640).all()
    b = shard_bounds(9, 2)
    assert list(b) == [0, 4, 9]
    assert list(shard_bounds(3, 8)) == [0, 0, 0, 1, 1, 1, 2, 2, 3] or shard_bounds(3, 8)[-1] == 3


def test_gather_plan_offsets_for_any_world_root_and_shard_sizes():
    """smrt_dort_gather_plan (the arithmetic smrt_dort_gather runs on, dort_comm.hip): simulate its transfers for 2..8
    ranks, every root, unequal and EMPTY shards -- the gathered buffer must be the rank-ordered concatenation."""
    from smrt_amd._native import gather_plan

    rng = np.random.default_rng(7)
    for world in range(1, 9):
        for trial in range(6):
            counts = rng.integers(0, 7, world)
            if trial == 0:
                counts[:] = 3
            rows = [np.arange(c) + 100 * r for r, c in enumerate(counts)]
            want = np.concatenate(rows) if world else np.array([])
            for root in range(world):
                got = np.full(int(counts.sum()), -1)
                sends = {}
                for rank in range(world):
                    ops, own, total = gather_plan(world, root, rank, counts)
                    assert total == counts.sum()
                    if rank == root:
                        got[own:own + counts[root]] = rows[root]
                        recvs = ops
                    else:
                        assert len(ops) == (1 if counts[rank] > 0 else 0)
                        for peer, off, n in ops:
                            assert peer == root and off == 0 and n == counts[rank]
                            sends[rank] = rows[rank][:n]
                assert sorted(p for p, _, _ in recvs) == sorted(sends)   # every send has its receive: nobody blocks
                for peer, off, n in recvs:
                    got[off:off + n] = sends[peer]
                assert np.array_equal(got, want)
    with pytest.raises(Exception):
        gather_plan(2, 0, 0, [1, -1])


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
