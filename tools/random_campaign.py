#!/usr/bin/env python3
"""Randomised differential campaign (development aid, GPU): pools of random length, size, chemistry, threshold
and -- every third case -- skewed base composition (large tables, pairs without a complementary cell, many ties)
through the exact-planes call and the decision-only call, against the oracle.
usage: random_campaign.py [seed] [cases] [k_lo] [k_hi] [only: comma-separated case numbers, the others still draw their random numbers]"""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / 'open-msspe-design_amd')); sys.path.insert(0, str(ROOT / 'oracle'))
import numpy as np
import msspe_amd as m, pyoracle as o
eng = m.Engine(0); tabs = o.Tables()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 77)
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 24
klo = int(sys.argv[3]) if len(sys.argv) > 3 else 9
khi = int(sys.argv[4]) if len(sys.argv) > 4 else 16
only = {int(x) for x in sys.argv[5].split(",")} if len(sys.argv) > 5 else None
bad = 0
for it in range(cases):
    k = int(rng.integers(klo, khi + 1))
    n = int(rng.integers(1200, 2600) if k <= 16 else rng.integers(300, 700))
    kw = [{}, dict(temp_c=37.0), dict(temp_c=55.0, dv=0.0), dict(mv=200.0, dv=0.5, dntp=0.2), dict(temp_c=15.0, mv=1200.0, dv=0.0)][int(rng.integers(0, 5))]
    thr = float(rng.choice([-9000.0, -6000.0, -3500.0, -1500.0]))
    seed = int(rng.integers(1, 1 << 30))
    # skewed compositions now and then: large tables, many ties
    p = rng.dirichlet([0.6] * 4) if it % 3 == 0 else None
    if only is not None and it not in only:
        continue
    g = np.random.default_rng(seed)
    pool = np.frombuffer(b"ACGT", dtype=np.uint8)[g.choice(4, size=(n, k), p=p)]
    strs = m.synth.pool_strings(pool)
    out = eng.cross_dimer(strs, m.Chem.ntthal(**kw), thr, want_dg=True, want_tm=True)
    fast = eng.cross_dimer(strs, m.Chem.ntthal(**kw), thr, want_dg=False, want_tm=False)
    cnt, dg, cf, tt = o.pool_pairs(tabs, pool, o.ntthal_args(**kw), thr, want_t=True)
    bits = np.unpackbits(out["bitmap"].view(np.uint8), axis=1, bitorder="little")[:, :n].astype(bool)
    fbits = np.unpackbits(fast["bitmap"].view(np.uint8), axis=1, bitorder="little")[:, :n].astype(bool)
    ok = [np.array_equal(out["dg"], dg), np.array_equal(out["tm"], tt), np.array_equal(bits, cf.astype(bool)), np.array_equal(fbits, cf.astype(bool))]
    print(it, "k", k, "n", n, kw, thr, "skew" if p is not None else "", "conflicts %.1f%%" % (100.0 * cnt / n / n), ok, flush=True)
    if not all(ok):   # the first few differing pairs, with both sides' numbers
        for name, got, want in (("dg", out["dg"], dg), ("tm", out["tm"], tt)):
            diff = np.argwhere(~((got == want) | (np.isnan(got) & np.isnan(want))))
            print("  ", name, "differs at", len(diff), "pairs")
            for r, c in diff[:6]:
                print("     ", int(r), int(c), strs[r], strs[c], "gpu", repr(float(got[r, c])), "oracle", repr(float(want[r, c])), flush=True)
        print("   stage stats:", eng.pair_stage_stats(), flush=True)
    bad += not all(ok)
print("BAD", bad)
sys.exit(1 if bad else 0)
