"""Multi-GPU tiling of the cross-dimer pair matrix, one process per GPU.

The O(N^2) pair matrix is cut into row blocks: rank r owns candidates [r0, r1) (the ones it
produced in stage B) and evaluates them against ALL columns, both orders of every pair being
covered because every ordered pair (i, j) belongs to exactly one row block.  Two collectives per
screening round, both a few MB even at 1M candidates:
    all_gather_into_tensor   packed pool shards  -> full packed pool on every rank
    all_reduce(sum)          per-primer conflict counts (each rank contributes its rows)
The conflict bitmap stays sharded by rows.  Works with backend "nccl" (= RCCL over xGMI) on GPUs
and with "gloo" (used by the tests to check the tiling logic itself: CPU tensors directly, device
tensors staged through the host, which lets two ranks rehearse the path on one GPU).
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_bounds(n: int, world: int, rank: int) -> tuple[int, int]:
    """Contiguous row block of `rank`; blocks differ by at most one row and cover [0, n)."""
    base, extra = divmod(n, world)
    r0 = rank * base + min(rank, extra)
    return r0, r0 + base + (1 if rank < extra else 0)


def dealt_rows(n: int, world: int, rank: int):
    """The pool rows rank `rank` screens when the POOL'S ORDER CARRIES MEANING (stage A emits its winners by falling
    frequency, and a row's cost depends on its base composition): groups of 256 rows dealt round robin -- the rule of
    include/msspe_hip.h msspe_group_rows, one rule for the in-process device group and for the one-process-per-GPU
    pipeline.  (bench.py's pools are uniformly random 13-mers: any cut is a fair sample there, and contiguous blocks
    keep the row range a single engine call -- shard_bounds.)  uint32 numpy array, ascending."""
    from .capi import group_rows
    return group_rows(n, world, rank)


def screen_dealt_rows_edges(engine, d_pool: torch.Tensor, rows, k: int, chem, threshold: float, capacity: int):
    """Conflict edges of the scattered row set `rows` (dealt_rows) against all columns of the pool: the rows are
    appended to a copy of the pool, P' = [pool | rows], and the engine screens the contiguous block [n, n + m) x
    [0, n) of P' (the layout csrc/group.hip uses); edge.a is mapped back through `rows`.  Returns (edges as a list
    of (a, b), count) -- count > capacity means the list was truncated (call again with more room)."""
    import numpy as np
    n, m = d_pool.numel(), int(len(rows))
    if m == 0:
        return [], 0
    dev = d_pool.device
    idx = torch.from_numpy(np.asarray(rows, dtype=np.int64)).to(dev)
    d_ext = torch.cat([d_pool, d_pool[idx]])
    d_edges = torch.zeros(capacity * 2, dtype=torch.int64, device=dev)       # 16-byte records
    d_count = torch.zeros(1, dtype=torch.int64, device=dev)
    engine.set_stream(torch.cuda.current_stream().cuda_stream)
    engine.cross_dimer_edges_dev(d_ext.data_ptr(), n + m, k, chem, threshold, (n, n + m), (0, n), d_edges.data_ptr(),
                                 capacity, d_count.data_ptr())
    torch.cuda.synchronize()
    engine.reset_stream()
    count = int(d_count.item())
    if count > capacity:
        return [], count
    rec = d_edges[: 2 * count].cpu().numpy().view(np.dtype([("a", np.uint32), ("b", np.uint32), ("dg", np.float64)]))
    return [(int(rows[int(e["a"]) - n]), int(e["b"])) for e in rec], count


def _staged(t: torch.Tensor) -> bool:
    """gloo has no device collectives here: stage device tensors through the host."""
    return t.is_cuda and dist.get_backend() == "gloo"


def all_reduce(t: torch.Tensor, op=dist.ReduceOp.SUM) -> torch.Tensor:
    """In-place all-reduce that also works for device tensors under gloo."""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return t
    if _staged(t):
        h = t.cpu()
        dist.all_reduce(h, op=op)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=op)
    return t


def gather_pool(local_shard: torch.Tensor, n: int) -> torch.Tensor:
    """All-gather the packed (int64) candidate shards into the full pool of n primers.
    Shards may differ in length by one: they are padded to the longest for the collective."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    if world == 1:
        return local_shard.clone()
    longest = -(-n // world)
    padded = torch.zeros(longest, dtype=local_shard.dtype, device=local_shard.device)
    padded[: local_shard.numel()] = local_shard
    if _staged(padded):
        host = torch.empty(longest * world, dtype=local_shard.dtype)
        dist.all_gather_into_tensor(host, padded.cpu())
        out = host.to(local_shard.device)
    else:
        out = torch.empty(longest * world, dtype=local_shard.dtype, device=local_shard.device)
        dist.all_gather_into_tensor(out, padded)
    parts = []
    for r in range(world):
        a, b = shard_bounds(n, world, r)
        parts.append(out[r * longest: r * longest + (b - a)])
    return torch.cat(parts)


def reduce_counts(local_counts: torch.Tensor) -> torch.Tensor:
    """Sum the per-primer conflict counts of all ranks in place (each rank filled its rows)."""
    return all_reduce(local_counts, dist.ReduceOp.SUM)


def screen_row_block(engine, d_pool: torch.Tensor, k: int, chem, threshold: float,
                     rows: tuple[int, int], d_counts: torch.Tensor, d_bitmap: torch.Tensor | None):
    """Launch the engine on this rank's row block (device tensors; asynchronous)."""
    n = d_pool.numel()
    engine.cross_dimer_dev(d_pool.data_ptr(), n, k, chem, threshold, rows, (0, n),
                           d_counts.data_ptr(), d_bitmap.data_ptr() if d_bitmap is not None else 0)
