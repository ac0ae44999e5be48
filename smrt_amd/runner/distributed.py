"""Multi-process runs, one rank per GPU: sharding of the flattened pair list and the single collective of the path --
the gather of the result rows to the root rank.

The pair list is embarrassingly parallel (every Model.run_single_simulation is independent, smrt/core/model.py:395-398;
the reference fans it out with joblib, smrt/runner/joblib_runner.py:45-72), so no other communication exists.

The product path needs no PyTorch: the gather is `smrt_dort_gather` of the C ABI (RCCL over xGMI, device buffer to
device buffer) and the ranks find each other with the few lines of TCP below -- rank 0 creates the RCCL id and serves it
on MASTER_ADDR at a port derived from MASTER_PORT (the variables torchrun / any launcher exports).  The CPU tests of the
sharding logic drive gloo themselves (tests/test_multirank_cpu.py) on `smrt_dort_gather_plan`."""
import hashlib
import os
import socket
import struct
import time

import numpy as np

_MAGIC = b"SMRTDORT"
_PORT_SPAN = 24   # candidate ports after MASTER_PORT


def shard_bounds(n_items, world_size):
    """Contiguous, count-balanced slices: rank r owns [bounds[r], bounds[r+1])."""
    return np.linspace(0, n_items, world_size + 1).astype(np.int64)


def _token():
    """What both sides must agree on before a payload is handed over (keeps stale servers and other jobs apart).  The
    launcher's variables are not secret: on a network where strangers can reach MASTER_ADDR, export the same random
    SMRT_DORT_JOB_SECRET to every rank (the launcher's environment) and it becomes part of the token -- a peer that does
    not know it is never handed the RCCL id and never counts as a rank."""
    key = "|".join(os.environ.get(k, "") for k in ("MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID", "WORLD_SIZE",
                                                    "SMRT_DORT_JOB_SECRET"))
    return hashlib.sha256(key.encode()).digest()[:16]


def _recv_exact(conn, n):
    buf = b""
    while len(buf) < n:
        chunk = conn.recv(n - len(buf))
        if not chunk:
            raise ConnectionError("peer closed the connection")
        buf += chunk
    return buf


def broadcast_from_root(payload, rank=None, world=None, addr=None, port=None, timeout=180.0):
    """Rank 0 hands `payload` (bytes) to every other rank; returns the payload on every rank.  Plain TCP on
    addr:(port + 1 ... port + 24) -- the first free one on the root, found by the others through the handshake."""
    rank = int(os.environ.get("RANK", "0")) if rank is None else rank
    world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else world
    addr = addr or os.environ.get("MASTER_ADDR", "127.0.0.1")
    port = int(os.environ.get("MASTER_PORT", "29500")) if port is None else int(port)
    if world == 1:
        return payload
    token = _token()
    deadline = time.time() + timeout
    if rank == 0:
        server = None
        for k in range(1, _PORT_SPAN + 1):
            s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            s.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            try:
                s.bind((addr, port + k))
                s.listen(world)
                server = s
                break
            except OSError:
                s.close()
        if server is None:
            raise RuntimeError("no free port next to MASTER_PORT for the rank rendezvous")
        served = set()
        server.settimeout(1.0)
        while len(served) < world - 1:
            if time.time() > deadline:
                raise TimeoutError("rank rendezvous: %d of %d ranks showed up" % (len(served) + 1, world))
            try:
                conn, _ = server.accept()
            except socket.timeout:
                continue
            with conn:
                conn.settimeout(10.0)
                try:
                    hello = _recv_exact(conn, len(_MAGIC) + 16 + 4)
                    peer = struct.unpack("<i", hello[-4:])[0]
                    if hello[:len(_MAGIC)] != _MAGIC or hello[len(_MAGIC):-4] != token or not 0 < peer < world:
                        continue
                    conn.sendall(_MAGIC + struct.pack("<q", len(payload)) + payload)
                    served.add(peer)
                except (OSError, ConnectionError):
                    continue
        server.close()
        return payload
    hello = _MAGIC + token + struct.pack("<i", rank)
    while time.time() < deadline:
        for k in range(1, _PORT_SPAN + 1):
            try:
                with socket.create_connection((addr, port + k), timeout=2.0) as conn:
                    conn.settimeout(10.0)
                    conn.sendall(hello)
                    head = _recv_exact(conn, len(_MAGIC) + 8)
                    if head[:len(_MAGIC)] != _MAGIC:
                        continue
                    return _recv_exact(conn, struct.unpack("<q", head[len(_MAGIC):])[0])
            except (OSError, ConnectionError):
                continue
        time.sleep(0.2)
    raise TimeoutError("rank %d could not reach rank 0 for the rendezvous" % rank)


def init_comm(ctx, rank=None, world=None):
    """Make `ctx` (a DortContext on this rank's GPU) a rank of the RCCL communicator of the job."""
    from .._native import DortContext

    rank = int(os.environ.get("RANK", "0")) if rank is None else rank
    world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else world
    uid = DortContext.comm_unique_id() if rank == 0 else None
    uid = broadcast_from_root(uid, rank, world)
    ctx.comm_init(world, rank, uid)
    return rank, world
