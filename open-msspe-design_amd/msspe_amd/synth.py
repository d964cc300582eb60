"""Synthetic inputs of BASELINE.md section 4 (seeded, reproducible)."""
from __future__ import annotations

import numpy as np

POOL_SEED = 20260630
_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def random_pool(n: int, k: int = 13, seed: int = POOL_SEED) -> np.ndarray:
    """n uniformly random k-mers over ACGT (PCG64), duplicates kept: uint8 ASCII (n, k)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    return _ACGT[rng.integers(0, 4, size=(n, k), dtype=np.uint8)]


def pool_strings(pool: np.ndarray) -> list[str]:
    return [bytes(r).decode() for r in pool]


def aligned_genomes(n_rows: int, length: int = 30000, seed: int = 1, sub_rate: float = 0.02,
                    gap_rate: float = 0.005, n_rate: float = 0.003) -> np.ndarray:
    """One random ancestor (seed) and n_rows descendants: 2 % substitutions, 0.5 % of columns
    turned to '-' in runs of 1-30, 0.3 % 'N' in runs of 1-200 (seed 1000+i).  uint8 (n_rows, length)."""
    anc = _ACGT[np.random.Generator(np.random.PCG64(seed)).integers(0, 4, size=length, dtype=np.uint8)]
    out = np.empty((n_rows, length), dtype=np.uint8)
    for i in range(n_rows):
        rng = np.random.Generator(np.random.PCG64(1000 + i))
        row = anc.copy()
        subs = rng.random(length) < sub_rate
        row[subs] = _ACGT[rng.integers(0, 4, size=int(subs.sum()), dtype=np.uint8)]
        for char, rate, max_run in ((ord("-"), gap_rate, 30), (ord("N"), n_rate, 200)):
            target = int(rate * length)
            done = 0
            while done < target:
                run = int(rng.integers(1, max_run + 1))
                run = min(run, target - done)
                pos = int(rng.integers(0, length - run + 1))
                row[pos:pos + run] = char
                done += run
        out[i] = row
    return out
