#!/usr/bin/env python3
"""GPU timing of stage A (k-mer candidates) on synthetic alignments (development aid)."""
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "open-msspe-design_amd"))
import numpy as np
import torch
import msspe_amd as m

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
length = int(sys.argv[2]) if len(sys.argv) > 2 else 30000
t0 = time.time()
g = m.synth.aligned_genomes(rows, length)
print(f"generated {rows} x {length} in {time.time()-t0:.1f} s", flush=True)
if os.environ.get("MSSPE_PROBE_LIB"):
    from msspe_amd import capi
    capi.use_library(os.environ["MSSPE_PROBE_LIB"])
eng = m.Engine(0)
for kv in os.environ.get("MSSPE_PROBE_OPTIONS", "").split(","):
    if kv:
        eng.set_option(*kv.split("="))
packed = os.environ.get("MSSPE_PROBE_ASCII", "") == ""
d = eng.put_rows_packed(g) if packed else torch.from_numpy(g).cuda()
opt = m.KmerOpt(500, 250, 50, 13, 1000, max(1, min(10, -(-rows // 50))))
for direction in (0, 1):
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.time()
        if packed:
            words, freqs = eng.kmer_candidates_packed(d, rows, length, opt, direction)
        else:
            words, freqs = eng.kmer_candidates(None, opt, direction, device_ptr=d.data_ptr(), n_seq=rows, seq_len=length)
        dt = time.time() - t0
    segs = rows * ((length - 500) // 250 + 1)
    its = [eng.info("stage_a_" + x) for x in ("fast_iterations", "general_iterations", "rebuilds", "idle_iterations")]
    tr = eng.kmer_trace()
    per_it = np.bincount(tr[:, 0]) if len(tr) else np.zeros(1, int)
    kinds = np.bincount(tr[:, 1], minlength=5).tolist() if len(tr) else []
    ends = [int(tr[np.flatnonzero(tr[:, 0] == i)[-1], 1]) for i in range(1, len(per_it)) if per_it[i]]
    print("   winners per iteration:", per_it[1:].tolist())
    print("   selected as [all-words, leader, several-partition, re-keyed, walked]:", kinds, " kind of each iteration's last winner:", np.bincount(ends, minlength=5).tolist())
    print(f"dir {direction}: {len(words)} winners, top freq {freqs[:3].tolist()}, {dt*1e3:.1f} ms, "
          f"{segs/dt/1e6:.2f} M segments/s; iterations fast/general/rebuilds/idle {its}", flush=True)
