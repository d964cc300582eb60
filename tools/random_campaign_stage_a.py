#!/usr/bin/env python3
"""Randomised differential campaign for stage A (development aid, GPU): alignments of random shape, divergence, gap
and N content, with random segment / stride / window / word length / iteration cap / frequency floor, through
both greedy-loop drivers, both directions, against the oracle.  usage: random_campaign_stage_a.py [seed] [cases]"""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "open-msspe-design_amd"))
sys.path.insert(0, str(ROOT / "oracle"))
import numpy as np
import msspe_amd as m
import pyoracle as o

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 11)
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 40
eng = m.Engine(0)
bad = 0
for it in range(cases):
    rows = int(rng.integers(2, 260))
    L = int(rng.integers(300, 14000))
    k = int(rng.integers(3, 21))
    win = int(rng.integers(k, k + 50))
    seg = int(rng.integers(win, win + 500))
    stride = int(rng.integers(win, win + 300))       # the reference refuses stride < window
    iters = int(rng.choice([1, 7, 60, 1000]))
    mm = int(rng.choice([1, 1, 2, 5]))
    mu = float(rng.choice([0.0, 0.002, 0.02, 0.15]))
    anc = rng.integers(0, 4, L)
    arr = np.empty((rows, L), dtype=np.uint8)
    for r in range(rows):
        row = anc.copy()
        mut = rng.random(L) < mu
        row[mut] = rng.integers(0, 4, int(mut.sum()))
        a = np.frombuffer(b"ACGT", dtype=np.uint8)[row].copy()
        if it % 2:                                   # gaps, N runs, ragged ends
            for _ in range(int(rng.integers(0, 4))):
                p0 = int(rng.integers(0, L)); a[p0:p0 + int(rng.integers(1, 40))] = ord("-")
            if rng.random() < 0.3:
                p0 = int(rng.integers(0, L)); a[p0:p0 + int(rng.integers(1, 15))] = ord("N")
            if rng.random() < 0.3:
                a[L - int(rng.integers(1, 200)):] = ord("-")
        arr[r] = a
    seqs = [bytes(r).decode() for r in arr]
    segs = o.Segments(seqs, seg, stride, win, k)
    ok = True
    for d in (0, 1):
        want = segs.candidates(d, iters, mm)
        for cand in (1, 0):
            eng.set_option("stage_a_candidates", cand)
            try:
                words, freqs = eng.kmer_candidates(arr, m.KmerOpt(seg, stride, win, k, iters, mm), d)
            finally:
                eng.set_option("stage_a_candidates", 1)
            ok = ok and list(zip(words, freqs.tolist())) == want
    print(it, f"rows {rows} L {L} k {k} win {win} seg {seg} stride {stride} iters {iters} mm {mm} mu {mu}", "gaps" if it % 2 else "",
          f"winners {len(want)}", ok, flush=True)
    bad += not ok
print("BAD", bad)
sys.exit(1 if bad else 0)
