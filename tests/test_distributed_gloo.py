"""The N > 1 tiling (row blocks + all-gather of candidate shards + all-reduce of conflict
counts) on CPU with the gloo backend, world_size 2.  The per-block compute is stood in for by the
oracle here (the GPU kernel is covered by the -m gpu tests); what is checked is that the sharded
result equals the single-process result."""
import os
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent


def _worker(rank: int, world: int, port: int, n: int, out_dir: str):
    for p in (ROOT / "open-msspe-design_amd", ROOT / "oracle"):
        sys.path.insert(0, str(p))
    import msspe_amd
    import pyoracle
    from msspe_amd.distributed import gather_pool, reduce_counts, shard_bounds

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pool_ascii = msspe_amd.synth.random_pool(n, 13, seed=31)
        packed = msspe_amd.pack_oligos(pool_ascii).view(np.int64)
        r0, r1 = shard_bounds(n, world, rank)
        full = gather_pool(torch.from_numpy(packed[r0:r1].copy()), n)
        assert torch.equal(full, torch.from_numpy(packed.copy()))
        # this rank's row block against all columns (oracle as the stand-in compute)
        tables = pyoracle.Tables()
        unpacked = [msspe_amd.unpack_oligo(int(w), 13) for w in full.numpy().view(np.uint64)]
        _, _, cf, _ = pyoracle.pool_pairs(tables, unpacked, rows=(r0, r1), threads=1, want_dg=False)
        counts = torch.zeros(n, dtype=torch.int32)
        counts[r0:r1] = torch.from_numpy(cf.sum(1).astype(np.int32))
        reduce_counts(counts)
        np.save(os.path.join(out_dir, f"counts_{rank}.npy"), counts.numpy())
        np.save(os.path.join(out_dir, f"bitmap_{rank}.npy"), cf)
    finally:
        dist.destroy_process_group()


def test_shard_bounds_cover_everything():
    from msspe_amd.distributed import shard_bounds
    for n in (0, 1, 7, 64, 1000, 65536):
        for world in (1, 2, 3, 8):
            blocks = [shard_bounds(n, world, r) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in blocks]
            assert max(sizes) - min(sizes) <= 1


def test_world_size_2_matches_single_process(tmp_path, oracle, oracle_tables):
    import msspe_amd
    n, world, port = 45, 2, 29500 + os.getpid() % 1000      # odd n: unequal shards
    mp.start_processes(_worker, args=(world, port, n, str(tmp_path)), nprocs=world, join=True,
                       start_method="spawn")
    pool = msspe_amd.synth.pool_strings(msspe_amd.synth.random_pool(n, 13, seed=31))
    _, _, cf, _ = oracle.pool_pairs(oracle_tables, pool, want_dg=False)
    for r in range(world):
        np.testing.assert_array_equal(np.load(tmp_path / f"counts_{r}.npy"), cf.sum(1))
    rows = np.concatenate([np.load(tmp_path / f"bitmap_{r}.npy") for r in range(world)])
    np.testing.assert_array_equal(rows, cf)


def _worker_dealt(rank: int, world: int, port: int, n: int, out_dir: str):
    for p in (ROOT / "open-msspe-design_amd", ROOT / "oracle"):
        sys.path.insert(0, str(p))
    import msspe_amd
    import pyoracle
    from msspe_amd.distributed import dealt_rows

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pool = msspe_amd.synth.pool_strings(msspe_amd.synth.random_pool(n, 13, seed=37))
        rows = dealt_rows(n, world, rank)
        tables = pyoracle.Tables()
        mine = []
        # the oracle as the stand-in compute, one call per run of consecutive rows (groups of 256)
        start = 0
        while start < len(rows):
            end = start
            while end + 1 < len(rows) and rows[end + 1] == rows[end] + 1:
                end += 1
            r0, r1 = int(rows[start]), int(rows[end]) + 1
            _, _, cf, _ = pyoracle.pool_pairs(tables, pool, rows=(r0, r1), threads=2, want_dg=False)
            mine += [(r0 + int(i), int(j)) for i, j in zip(*np.nonzero(cf))]
            start = end + 1
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
        if rank == 0:
            edges = sorted(e for part in gathered for e in part)
            np.save(os.path.join(out_dir, "edges.npy"), np.array(edges, dtype=np.int64).reshape(-1, 2))
            np.save(os.path.join(out_dir, "sizes.npy"), np.array([len(p) for p in gathered]))
    finally:
        dist.destroy_process_group()


def test_dealt_rows_world_size_2_matches_single_process(tmp_path, oracle, oracle_tables):
    """The pipeline's row dealing (groups of 256, round robin: distributed.dealt_rows = msspe_group_rows): two ranks'
    edge lists, gathered and sorted by (a, b), are the single-process edge list."""
    import msspe_amd
    n, world, port = 600, 2, 30500 + os.getpid() % 1000      # rank 0: rows 0..255 and 512..599, rank 1: 256..511
    mp.start_processes(_worker_dealt, args=(world, port, n, str(tmp_path)), nprocs=world, join=True, start_method="spawn")
    pool = msspe_amd.synth.pool_strings(msspe_amd.synth.random_pool(n, 13, seed=37))
    _, _, cf, _ = oracle.pool_pairs(oracle_tables, pool, want_dg=False)
    want = np.array(sorted((int(i), int(j)) for i, j in zip(*np.nonzero(cf))), dtype=np.int64).reshape(-1, 2)
    np.testing.assert_array_equal(np.load(tmp_path / "edges.npy"), want)
    assert len(want) > 100 and np.load(tmp_path / "sizes.npy").min() > 0
