#!/usr/bin/env python3
"""Times msspe_oligo_stats (stage B: Tm, GC %, SELF_ANY / SELF_END / HAIRPIN) for n oligos."""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "open-msspe-design_amd"))
import msspe_amd as m

n = int(sys.argv[1]) if len(sys.argv) > 1 else 700
k = int(sys.argv[2]) if len(sys.argv) > 2 else 13
eng = m.Engine(0)
pool = m.synth.pool_strings(m.synth.random_pool(n, k, seed=3))
eng.oligo_stats(pool)
t0 = time.time()
for _ in range(5):
    eng.oligo_stats(pool)
print(f"oligo_stats n={n} k={k}: {(time.time() - t0) / 5 * 1e3:.2f} ms per call", flush=True)
