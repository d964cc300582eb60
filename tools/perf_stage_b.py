#!/usr/bin/env python3
"""Times msspe_oligo_stats_dev (stage B: Tm, GC %, SELF_ANY / SELF_END / HAIRPIN) part by part for n oligos."""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "open-msspe-design_amd"))
import numpy as np
import torch
import msspe_amd as m

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
k = int(sys.argv[2]) if len(sys.argv) > 2 else 13
eng = m.Engine(0)
eng.set_stream(torch.cuda.current_stream().cuda_stream)
import os
for kv in os.environ.get("MSSPE_PROBE_OPTIONS", "").split(","):
    if kv:
        eng.set_option(*kv.split("="))
pool = m.synth.random_pool(n, k, seed=2000 + n)
d_pool = torch.from_numpy(m.pack_oligos(pool).view(np.int64)).cuda()
d_out = torch.zeros((5, n), dtype=torch.float64, device="cuda")
ptr = [d_out[q].data_ptr() for q in range(5)]
parts = {"tm_gc": (ptr[0], ptr[1], 0, 0, 0), "self_any": (0, 0, ptr[2], 0, 0), "self_end": (0, 0, 0, ptr[3], 0),
         "self_dimers": (0, 0, ptr[2], ptr[3], 0), "hairpin": (0, 0, 0, 0, ptr[4]), "all": tuple(ptr)}
chem = m.Chem.primer3()
for name, a in parts.items():
    eng.oligo_stats_dev(d_pool.data_ptr(), n, k, chem, *a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        eng.oligo_stats_dev(d_pool.data_ptr(), n, k, chem, *a)
    e1.record()
    torch.cuda.synchronize()
    print(f"{name:11s} n={n} k={k}: {e0.elapsed_time(e1) / 3:.3f} ms", flush=True)
print("hairpin > 0:", int((d_out[4] > 0).sum().item()), "of", n)
