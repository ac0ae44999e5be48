"""Multi-process (one rank per GPU) helpers: contiguous sharding of the flattened pair list and the single collective
of the path -- the gather of the results to the root rank (RCCL over xGMI with backend "nccl", gloo in CPU tests).

The pair list is embarrassingly parallel (every Model.run_single_simulation is independent, smrt/core/model.py:395-398),
so no other communication exists."""
import numpy as np


def shard_bounds(n_items, world_size):
    """Contiguous, balanced slices: rank r owns [bounds[r], bounds[r+1])."""
    return np.linspace(0, n_items, world_size + 1).astype(np.int64)


def gather_to_root(dist, values, status, dst=0):
    """Gather per-rank result rows (torch tensors on the collective's device) to `dst`.  Shards may have different
    lengths: rows are padded to the longest shard for the collective and trimmed afterwards.
    Returns (values, status) concatenated in rank order on `dst`, (None, None) elsewhere."""
    import torch

    world, rank = dist.get_world_size(), dist.get_rank()
    n_local = torch.tensor([values.shape[0]], dtype=torch.int64, device=values.device)
    counts = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(counts, n_local)
    counts = [int(c.item()) for c in counts]
    n_max = max(counts)
    pad_v = torch.zeros((n_max,) + tuple(values.shape[1:]), dtype=values.dtype, device=values.device)
    pad_s = torch.zeros((n_max,), dtype=status.dtype, device=status.device)
    pad_v[: values.shape[0]] = values
    pad_s[: status.shape[0]] = status
    if rank == dst:
        gv = [torch.empty_like(pad_v) for _ in range(world)]
        gs = [torch.empty_like(pad_s) for _ in range(world)]
    else:
        gv = gs = None
    dist.gather(pad_v, gv, dst=dst)
    dist.gather(pad_s, gs, dst=dst)
    if rank != dst:
        return None, None
    return (torch.cat([g[:c] for g, c in zip(gv, counts)], dim=0), torch.cat([g[:c] for g, c in zip(gs, counts)], dim=0))
