#!/usr/bin/env python3
"""Randomised differential campaign for the device-pointer screen (development aid, GPU): random rectangular
blocks of the pair matrix with a random subset of the outputs (counts, bitmap, dG plane, Tm plane, edge list) on
pre-filled buffers, against the oracle -- what a rank of the multi-GPU tiling and the C++ host ask for.
usage: random_campaign_blocks.py [seed] [cases]"""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "open-msspe-design_amd"))
sys.path.insert(0, str(ROOT / "oracle"))
import numpy as np
import torch
import msspe_amd as m
import pyoracle as o

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 8)
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 40
eng = m.Engine(0)
tabs = o.Tables()
bad = 0
for it in range(cases):
    k = int(rng.choice([9, 11, 12, 13, 13, 13, 14, 16, 20]))
    n = int(rng.integers(100, 2200 if k <= 16 else 500))
    thr = float(rng.choice([-9000.0, -5000.0, -2500.0]))
    kw = [{}, dict(temp_c=37.0, dv=1.5, dntp=0.6, dna_conc=50.0), dict(temp_c=50.0)][int(rng.integers(0, 3))]
    p = rng.dirichlet([0.7] * 4) if it % 4 == 0 else None
    pool = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.choice(4, size=(n, k), p=p)]
    r0 = int(rng.integers(0, n)); r1 = int(rng.integers(r0 + 1, n + 1))
    c0 = int(rng.integers(0, n)); c1 = int(rng.integers(c0 + 1, n + 1))
    want = {x: bool(rng.integers(0, 2)) for x in ("rc", "bm", "dg", "tm")}
    edges = not (want["bm"] or want["dg"] or want["tm"]) and bool(rng.integers(0, 2))
    if not any(want.values()) and not edges:
        want["rc"] = True
    d_pool = torch.from_numpy(m.pack_oligos(pool).view(np.int64)).cuda()
    R, Cc = r1 - r0, c1 - c0
    d_rc = torch.zeros(n, dtype=torch.int32, device="cuda")
    d_bm = torch.zeros((R, (Cc + 63) // 64), dtype=torch.int64, device="cuda")
    d_dg = torch.full((R, Cc), 7.0, dtype=torch.float64, device="cuda")
    d_tm = torch.full((R, Cc), 7.0, dtype=torch.float64, device="cuda")
    chem = m.Chem.ntthal(**kw)
    _, dg, cf, tt = o.pool_pairs(tabs, pool, o.ntthal_args(**kw), thr, rows=(r0, r1), want_t=True)
    dg, cf, tt = dg[:, c0:c1], cf[:, c0:c1], tt[:, c0:c1]
    ok = []
    if edges:
        cap = int(cf.sum()) + 8
        d_e = torch.zeros(cap * 2, dtype=torch.int64, device="cuda")
        d_cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
        eng.cross_dimer_edges_dev(d_pool.data_ptr(), n, k, chem, thr, (r0, r1), (c0, c1), d_e.data_ptr(), cap,
                                  d_cnt.data_ptr(), d_rc.data_ptr() if want["rc"] else 0)
        eng.synchronize()
        cnt = int(d_cnt.item())
        rec = d_e.cpu().numpy()[: 2 * cnt].reshape(-1, 2)
        ab = np.ascontiguousarray(rec[:, 0]).view(np.uint32).reshape(-1, 2)
        got = sorted((int(a), int(b), float(g)) for (a, b), g in zip(ab, np.ascontiguousarray(rec[:, 1]).view(np.float64)))
        ref = sorted((r0 + int(i), c0 + int(j), float(dg[i, j])) for i, j in np.argwhere(cf))
        ok.append(cnt == int(cf.sum()) and got == ref)
    else:
        eng.cross_dimer_dev(d_pool.data_ptr(), n, k, chem, thr, (r0, r1), (c0, c1), d_rc.data_ptr() if want["rc"] else 0,
                            d_bm.data_ptr() if want["bm"] else 0, d_dg.data_ptr() if want["dg"] else 0,
                            d_tm.data_ptr() if want["tm"] else 0)
        eng.synchronize()
    if want["rc"]:
        w = np.zeros(n, dtype=np.int64); w[r0:r1] = cf.sum(1)
        ok.append(np.array_equal(d_rc.cpu().numpy().astype(np.int64), w))
    if want["bm"] and not edges:
        bits = np.unpackbits(d_bm.cpu().numpy().view(np.uint8), axis=1, bitorder="little")[:, :Cc].astype(bool)
        ok.append(np.array_equal(bits, cf.astype(bool)))
    if want["dg"] and not edges:
        ok.append(np.array_equal(d_dg.cpu().numpy(), dg))
    if want["tm"] and not edges:
        ok.append(np.array_equal(d_tm.cpu().numpy(), tt))
    print(it, f"k {k} n {n} rows {r0}:{r1} cols {c0}:{c1} thr {thr} {kw}", "skew" if p is not None else "",
          "edges" if edges else [x for x in want if want[x]], f"conflicts {int(cf.sum())}", ok, flush=True)
    bad += not all(ok)
print("BAD", bad)
sys.exit(1 if bad else 0)
