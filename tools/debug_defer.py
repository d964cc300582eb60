#!/usr/bin/env python3
"""Development aid: which pairs does the integer stage hand on, and does the CPU model agree?"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "open-msspe-design_amd"))
sys.path.insert(0, str(ROOT / "oracle"))
sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
import torch
import msspe_amd as m
import int_dp_model as model

tb = model.load_tables(m)
eng = m.Engine(0)
chem = m.Chem.ntthal()
eng.set_stream(torch.cuda.current_stream().cuda_stream)


def screen(oligos):
    n = len(oligos)
    pool = m.pack_oligos(oligos)
    d_pool = torch.from_numpy(pool.view(np.int64)).cuda()
    d_rc = torch.zeros(n, dtype=torch.int32, device="cuda")
    d_bm = torch.zeros((n, (n + 63) // 64), dtype=torch.int64, device="cuda")
    eng.pair_stage_stats()
    eng.cross_dimer_dev(d_pool.data_ptr(), n, 13, chem, -9000.0, (0, n), (0, n), d_rc.data_ptr(), d_bm.data_ptr())
    torch.cuda.synchronize()
    samples = eng.pair_stage_samples()
    return samples, eng.pair_stage_stats()


def s(x):
    return x if isinstance(x, str) else bytes(x).decode()


big = [s(x) for x in m.synth.random_pool(256, 13)]
samples, stats = screen(big)
print(stats)
agree = 0
for row, col, bits in samples[:200]:
    _, d, _ = model.run_pair(tb, big[row], big[col])
    agree += (d != 0)
print("model also defers", agree, "of", min(200, len(samples)))
for n_small in (2, 3, 5, 64, 65):
    sm, st = screen(big[:n_small])
    want = sum(model.run_pair(tb, a, b)[1] != 0 for a in big[:n_small] for b in big[:n_small])
    print(n_small, "gpu deferred", st["deferred"], "model", want, st)
