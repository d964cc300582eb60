#!/usr/bin/env python3
"""Development aid (GPU): one case of tools/random_campaign.py again, under several engine options, printing chosen
pairs with the oracle's numbers beside them.  usage: debug_pair.py seed case k_lo k_hi row:col[,row:col...] [opt=val,...;opt=val...]"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / 'open-msspe-design_amd')); sys.path.insert(0, str(ROOT / 'oracle'))
import numpy as np
import msspe_amd as m, pyoracle as o
tabs = o.Tables()
rng = np.random.default_rng(int(sys.argv[1]))
case = int(sys.argv[2]); klo = int(sys.argv[3]); khi = int(sys.argv[4])
pairs = [tuple(int(x) for x in p.split(":")) for p in sys.argv[5].split(",")]
variants = sys.argv[6].split(";") if len(sys.argv) > 6 else [""]
for it in range(case + 1):
    k = int(rng.integers(klo, khi + 1))
    n = int(rng.integers(1200, 2600) if k <= 16 else rng.integers(300, 700))
    kw = [{}, dict(temp_c=37.0), dict(temp_c=55.0, dv=0.0), dict(mv=200.0, dv=0.5, dntp=0.2), dict(temp_c=15.0, mv=1200.0, dv=0.0)][int(rng.integers(0, 5))]
    thr = float(rng.choice([-9000.0, -6000.0, -3500.0, -1500.0]))
    seed = int(rng.integers(1, 1 << 30))
    p = rng.dirichlet([0.6] * 4) if it % 3 == 0 else None
g = np.random.default_rng(seed)
pool = np.frombuffer(b"ACGT", dtype=np.uint8)[g.choice(4, size=(n, k), p=p)]
strs = m.synth.pool_strings(pool)
print("case", case, "k", k, "n", n, kw, thr, flush=True)
cnt, dg, cf, tt = o.pool_pairs(tabs, pool, o.ntthal_args(**kw), thr, want_t=True)
for v in variants:
    eng = m.Engine(0)
    for kv in v.split(","):
        if kv:
            eng.set_option(*kv.split("="))
    out = eng.cross_dimer(strs, m.Chem.ntthal(**kw), thr, want_dg=True, want_tm=True)
    bad_dg = int((~((out["dg"] == dg) | (np.isnan(out["dg"]) & np.isnan(dg)))).sum())
    bad_tm = int((~((out["tm"] == tt) | (np.isnan(out["tm"]) & np.isnan(tt)))).sum())
    print(f"[{v or 'default'}] differing dg {bad_dg} tm {bad_tm}", flush=True)
    for r, c in pairs:
        print("    ", r, c, strs[r], strs[c], "gpu dg", repr(float(out["dg"][r, c])), "tm", repr(float(out["tm"][r, c])),
              "| oracle dg", repr(float(dg[r, c])), "tm", repr(float(tt[r, c])), flush=True)
    st = eng.pair_stage_stats()
    print("     stats:", {k2: v2 for k2, v2 in st.items() if not isinstance(v2, dict)}, flush=True)
    eng.close()
