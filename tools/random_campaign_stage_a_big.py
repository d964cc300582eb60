#!/usr/bin/env python3
"""Randomised differential campaign for stage A's multi-winner loop (development aid, GPU): larger alignments in the
regime the fast path is made for -- hundreds to thousands of near-identical rows, default windows -- with the things
that stress it: duplicated column blocks (words with many postings in SEVERAL partitions), clades (several
ancestors), gaps and N runs.  The candidate-list loop against the all-words loop (one winner per iteration: round 1's
loop, itself checked against the oracle by the other campaign), both directions, whole winner sequences with their
frequencies; AND against the oracle: through the committed hashes of the oracle's winner sequences where
tests/golden/stage_a_big_campaign.json holds the case (the fixed-seed cases the suite runs), else by running the oracle
beside the GPU where the case is affordable (every case this generator draws: rows x L <= 30,000,000, seconds each).
usage: random_campaign_stage_a_big.py [seed] [cases] [only]                 (GPU)
       random_campaign_stage_a_big.py --oracle-hashes seed cases [seed cases ...]   (build container, no GPU: writes
                                                                                    the hashes of every case)"""
import hashlib
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "open-msspe-design_amd"))
sys.path.insert(0, str(ROOT / "oracle"))
import numpy as np
import pyoracle as o

HASHES = ROOT / "tests" / "golden" / "stage_a_big_campaign.json"
ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def winners_hash(pairs) -> str:
    return hashlib.sha256(json.dumps([[w, int(f)] for w, f in pairs]).encode()).hexdigest()


def draw_case(rng, it):
    """Case `it` of the generator `rng` (every case draws its random numbers, evaluated or not)."""
    rows = int(rng.integers(150, 2500))
    L = int(rng.integers(2000, 12000))
    k = int(rng.choice([9, 11, 13, 13, 13, 15]))
    seg, stride, win = (500, 250, 50) if it % 3 else (int(rng.integers(200, 600)), int(rng.integers(60, 300)), int(rng.integers(k + 5, 60)))
    iters = int(rng.choice([1000, 1000, 300, 40]))
    mm = int(rng.choice([1, 2, 5, 10]))
    mu = float(rng.choice([0.004, 0.02, 0.02, 0.06]))
    clades = int(rng.choice([1, 1, 2, 4]))
    anc = [rng.integers(0, 4, L) for _ in range(clades)]
    if it % 2:   # duplicated blocks: the same words in two or three partitions
        for a in anc:
            for _ in range(int(rng.integers(1, 4))):
                w = int(rng.integers(30, 400))
                src, dst = int(rng.integers(0, L - w)), int(rng.integers(0, L - w))
                a[dst:dst + w] = a[src:src + w]
    arr = np.empty((rows, L), dtype=np.uint8)
    for r in range(rows):
        row = anc[r % clades].copy()
        mut = rng.random(L) < mu
        row[mut] = rng.integers(0, 4, int(mut.sum()))
        a = ACGT[row].copy()
        if it % 4 == 3:
            for _ in range(int(rng.integers(0, 3))):
                p0 = int(rng.integers(0, L)); a[p0:p0 + int(rng.integers(1, 60))] = ord("-")
            if rng.random() < 0.2:
                p0 = int(rng.integers(0, L)); a[p0:p0 + int(rng.integers(1, 150))] = ord("N")
        arr[r] = a
    return arr, (seg, stride, win, k, iters, mm), dict(rows=rows, L=L, mu=mu, clades=clades)


if len(sys.argv) > 1 and sys.argv[1] == "--oracle-hashes":
    doc = json.loads(HASHES.read_text()) if HASHES.exists() else {
        "provenance": "tools/random_campaign_stage_a_big.py --oracle-hashes: sha256 of json.dumps([[word, frequency], ...]) of "
                      "oracle/stage_a.c's find_candidates_kmers restatement (main.rs:331-406) per case and direction",
        "cases": {}}
    for seed, cases in zip(sys.argv[2::2], sys.argv[3::2]):
        rng = np.random.default_rng(int(seed))
        for it in range(int(cases)):
            arr, (seg, stride, win, k, iters, mm), _ = draw_case(rng, it)
            segs = o.Segments([bytes(r).decode() for r in arr], seg, stride, win, k)
            doc["cases"][f"{seed}:{it}"] = [winners_hash(segs.candidates(d, iters, mm)) for d in (0, 1)]
            print(seed, it, arr.shape, doc["cases"][f"{seed}:{it}"][0][:12], flush=True)
    HASHES.write_text(json.dumps(doc, indent=1) + "\n")
    sys.exit(0)

import msspe_amd as m

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 5
rng = np.random.default_rng(seed)
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 30
only = int(sys.argv[3]) if len(sys.argv) > 3 else -1    # evaluate this case alone (the others still draw their random numbers)
committed = json.loads(HASHES.read_text())["cases"] if HASHES.exists() else {}
eng = m.Engine(0)
bad = 0
n_hash = n_oracle = 0
for it in range(cases):
    arr, (seg, stride, win, k, iters, mm), meta = draw_case(rng, it)
    rows, L, mu, clades = meta["rows"], meta["L"], meta["mu"], meta["clades"]
    opt = m.KmerOpt(seg, stride, win, k, iters, mm)
    if only >= 0 and it != only:
        continue
    ok = True
    info = []
    for d in (0, 1):
        eng.set_option("stage_a_candidates", 0)
        w0, f0 = eng.kmer_candidates(arr, opt, d)
        eng.set_option("stage_a_candidates", 1)
        w1, f1 = eng.kmer_candidates(arr, opt, d)
        tr = eng.kmer_trace()
        same = list(w0) == list(w1) and f0.tolist() == f1.tolist()
        if not same and only >= 0:
            n = min(len(w0), len(w1))
            b = next((i for i in range(n) if w0[i] != w1[i] or f0[i] != f1[i]), n)
            print(f"dir {d}: {len(w0)} vs {len(w1)} winners, first difference at {b}")
            for i in range(max(0, b - 6), min(n, b + 4)):
                print("  ", i, w0[i], int(f0[i]), "|", w1[i], int(f1[i]), tr[i].tolist())
            # where the words around the difference live: partition -> segments holding them (all of them, covered or not)
            P = (L - seg) // stride + 1
            comp = str.maketrans("ACGT", "TGCA")
            for word in sorted({w0[b], w1[b], w0[b - 1]}):
                look = word if d == 0 else word[::-1].translate(comp)
                cnt = {}
                for r in range(rows):
                    row = bytes(arr[r]).decode()
                    for part in range(P):
                        c0 = part * stride + (seg - win if d else 0)
                        if look in row[c0:c0 + win]:
                            cnt[part] = cnt.get(part, 0) + 1
                print("   word", word, "partitions", cnt)
        key = f"{seed}:{it}"
        if key in committed and only < 0:
            # the oracle's winner sequence for this case, computed in the build container and committed
            same = same and winners_hash(zip(w1, f1.tolist())) == committed[key][d] == winners_hash(zip(w0, f0.tolist()))
            n_hash += 1
        elif (same and rows * L <= 40_000_000) or only >= 0:
            n_oracle += 1
            want = o.Segments([bytes(r).decode() for r in arr], seg, stride, win, k).candidates(d, iters, mm)
            if only >= 0:
                print("oracle == all-words loop:", list(zip(w0, f0.tolist())) == want, " oracle == candidate-list loop:",
                      list(zip(w1, f1.tolist())) == want)
            same = same and list(zip(w1, f1.tolist())) == want
        ok = ok and same
        hist = np.bincount(tr[:, 1], minlength=5).tolist() if len(tr) else [0] * 5
        info.append(f"{len(w1)} winners in {int(tr[:, 0].max()) if len(tr) else 0} it. {hist}")
    print(it, f"rows {rows} L {L} k {k} seg {seg}/{stride}/{win} iters {iters} mm {mm} mu {mu} clades {clades}", "dup" if it % 2 else "",
          "|", " ; ".join(info), ok, flush=True)
    bad += not ok
print("BAD", bad, "| directions checked against committed oracle hashes:", n_hash, "against the oracle run here:", n_oracle)
sys.exit(1 if bad else 0)
