#!/usr/bin/env python3
"""Randomised differential campaign for stage B (development aid, GPU): oligo sets of random length (8 ... 32) and
base composition -- every third one skewed, every fourth one with designed stem-loops -- through msspe_oligo_stats
(Tm, GC %, SELF_ANY, SELF_END, HAIRPIN at primer3_core's settings) against the oracle, bit for bit.
usage: random_campaign_stage_b.py [seed] [cases]"""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "open-msspe-design_amd"))
sys.path.insert(0, str(ROOT / "oracle"))
import numpy as np
import msspe_amd as m
import pyoracle as o

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 3)
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 40
eng = m.Engine(0)
tabs = o.Tables()
comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
bad = 0
for it in range(cases):
    k = int(rng.integers(8, 33))
    n = int(rng.integers(50, 1500))
    p = rng.dirichlet([0.5] * 4) if it % 3 == 0 else None
    pool = ["".join("ACGT"[x] for x in rng.choice(4, size=k, p=p)) for _ in range(n)]
    if it % 4 == 0 and k >= 12:                      # stem-loops: stem + loop + reverse complement of the stem
        for j in range(0, n, 3):
            stem = int(rng.integers(3, (k - 3) // 2 + 1))
            s = pool[j][:stem]
            loop = pool[j][stem:k - stem]
            pool[j] = s + loop + "".join(comp[c] for c in reversed(s))
    # every other case: the self-dimers through the one-lane-per-oligo kernels of large pools (k <= 16; longer oligos
    # stay on the wave kernel whatever the option says)
    eng.set_option("self_lane_from", 0 if it % 2 else 81920)
    got = eng.oligo_stats(pool)
    ref = o.check_primers(tabs, pool)
    ok = [np.array_equal(got["tm"], ref["tm"]), np.array_equal(got["gc"], ref["gc"]),
          np.array_equal(got["self_any"], ref["self_any_th"]), np.array_equal(got["self_end"], ref["self_end_th"]),
          np.array_equal(got["hairpin"], ref["hairpin_th"])]
    print(it, "k", k, "n", n, "skew" if p is not None else "", "stems" if it % 4 == 0 and k >= 12 else "",
          "hairpins > 0:", int((ref["hairpin_th"] > 0).sum()), ok, flush=True)
    bad += not all(ok)
print("BAD", bad)
sys.exit(1 if bad else 0)
