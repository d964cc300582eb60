#!/usr/bin/env python3
"""One-off large check of the decision-only screen (bitmap + counts, no dG / Tm planes: tied picks and path ties
may stand, last rows are unstored, hand-over lists are flushed rarely) against the CPU oracle's decisions, at
thresholds inside the bulk of the dG distribution and for several chemistries; too slow on the CPU side for the
test suite (16.8 M pairs per case)."""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "open-msspe-design_amd"))
sys.path.insert(0, str(ROOT / "oracle"))
import numpy as np
import msspe_amd as m
import pyoracle as o

eng = m.Engine(0)
tabs = o.Tables()
cases = ((13, {}, -9000.0), (13, {}, -4000.0), (13, {}, -2000.0), (13, dict(temp_c=60.0, dv=0.0), -800.0),
         (13, dict(temp_c=37.0, mv=100.0, dv=1.5, dntp=0.2, dna_conc=50.0), -3000.0), (12, {}, -2500.0),
         (11, dict(temp_c=10.0, mv=1500.0, dv=0.0), -5000.0), (14, {}, -3000.0))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
for k, kw, thr in cases:
    pool = m.synth.random_pool(n, k, seed=4000 + k + int(-thr))
    t0 = time.time()
    out = eng.cross_dimer(m.synth.pool_strings(pool), m.Chem.ntthal(**kw), thr, want_dg=False, want_tm=False)
    t1 = time.time()
    cnt, _, cf, _ = o.pool_pairs(tabs, pool, o.ntthal_args(**kw), thr, want_dg=False)
    t2 = time.time()
    bits = np.unpackbits(out["bitmap"].view(np.uint8), axis=1, bitorder="little")[:, :n].astype(bool)
    ok = np.array_equal(bits, cf.astype(bool)) and np.array_equal(out["row_conflicts"].astype(np.int64), cf.sum(1))
    print(f"k={k} n={n} {kw} thr={thr}: decisions equal {ok} ({cnt} conflicts = {cnt / n / n:.1%}), gpu {t1 - t0:.2f} s, "
          f"oracle {t2 - t1:.1f} s", flush=True)
    assert ok
print("ALL OK")
