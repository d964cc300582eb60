#!/usr/bin/env python3
"""Development aid: the candidate-list loop against the all-words loop on one case; first difference + trace."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "open-msspe-design_amd"))
import numpy as np
import msspe_amd as m
case = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cases = [(m.synth.aligned_genomes(400, 9000), m.KmerOpt(500, 250, 50, 13, 1000, 1)),
         (m.synth.aligned_genomes(120, 3000), m.KmerOpt(300, 100, 40, 6, 1000, 1)),
         (m.synth.aligned_genomes(900, 30000), m.KmerOpt(500, 250, 50, 13, 300, 3))]
arr, opt = cases[case]
eng = m.Engine(0)
for d in (0, 1):
    eng.set_option("stage_a_candidates", 0)
    w0, f0 = eng.kmer_candidates(arr, opt, d)
    eng.set_option("stage_a_candidates", 1)
    w1, f1 = eng.kmer_candidates(arr, opt, d)
    tr = eng.kmer_trace()
    n = min(len(w0), len(w1))
    bad = [i for i in range(n) if w0[i] != w1[i] or f0[i] != f1[i]]
    print(f"dir {d}: {len(w0)} vs {len(w1)} winners; first difference at {bad[0] if bad else None}")
    if bad:
        b = bad[0]
        for i in range(max(0, b - 4), min(n, b + 4)):
            print(i, w0[i], int(f0[i]), "|", w1[i], int(f1[i]), tr[i].tolist())
    hist = np.bincount(tr[:, 1], minlength=5)
    print("selected as [all-words, leader, several-partition, re-keyed, walked]:", hist.tolist(), "iterations", int(tr[:, 0].max()) if len(tr) else 0)
