#!/usr/bin/env python3
"""Writes tests/golden/config2_10k.json: BASELINE.json configs[2] (10,000 synthetic aligned 30 kb genomes) through the
CPU oracle at full size.  Run once in the build container (no GPU; minutes, ~10 GB of memory):

    python tools/make_config2_fixture.py [--rows 10000] [--length 30000] [--out tests/golden/config2_10k.json]

What is kept (data only -- inputs are regenerated from msspe_amd.synth's seeds by whoever reads the fixture):
  * both directions' whole winner sequences with their frequencies, from oracle/stage_a.c's restatement of
    find_candidates_kmers (/root/reference/od-msspe/src/main.rs:331-406) at the reference's default options
    (segment 500, stride 250, window 50, k = 13, max_iterations 1000, max_mismatch_segments = the automatic
    rule of main.rs:658-660);
  * the primers that survive the default filters and the vertex cover, per direction;
  * sha256 of the CSV text and the coverage report text of oracle/ref_pipeline.run (main.rs:596-861 restated).
The FASTA the hashes belong to is ">genome{i}\\n{row}\\n" per row (tests/test_gpu_baseline_configs.py writes the same).
"""
from __future__ import annotations

import argparse
import hashlib
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "open-msspe-design_amd"))
sys.path.insert(0, str(ROOT / "oracle"))


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=10000)
    ap.add_argument("--length", type=int, default=30000)
    ap.add_argument("--out", default=str(ROOT / "tests" / "golden" / "config2_10k.json"))
    a = ap.parse_args()

    import pyoracle
    import ref_pipeline
    from msspe_amd import synth

    t0 = time.time()
    g = synth.aligned_genomes(a.rows, a.length)
    rows = [bytes(r).decode() for r in g]
    print(f"genomes: {time.time() - t0:.1f} s", flush=True)
    mm = min(10, max(1, -(-a.rows // 50)))
    t0 = time.time()
    segs = pyoracle.Segments(rows, 500, 250, 50, 13)
    print(f"segments + index: {len(segs)} segments, {time.time() - t0:.1f} s", flush=True)
    cands = {}
    for d in (0, 1):
        t0 = time.time()
        cands[d] = segs.candidates(d, 1000, mm)
        print(f"direction {d}: {len(cands[d])} winners, {time.time() - t0:.1f} s", flush=True)
    del segs
    fasta = "".join(f">genome{i}\n{r}\n" for i, r in enumerate(rows))
    t0 = time.time()
    csv, report, info = ref_pipeline.run(fasta, candidates=cands)
    print(f"pipeline: {time.time() - t0:.1f} s", flush=True)
    kept = {"F": [], "R": []}
    for line in csv.splitlines()[1:]:
        f = line.split(",")
        kept[f[0]].append(f[2])
    doc = {
        "provenance": "tools/make_config2_fixture.py: oracle/stage_a.c + oracle/ref_pipeline.py on "
                      f"msspe_amd.synth.aligned_genomes({a.rows}, {a.length}) (seed 1, rows 1000+i), default options; "
                      "restates /root/reference/od-msspe/src/main.rs:331-406 and :596-861",
        "rows": a.rows, "length": a.length,
        "options": {"segment": 500, "stride": 250, "window": 50, "k": 13, "max_iterations": 1000,
                    "max_mismatch_segments": mm},
        "winners": {str(d): [[w, f] for w, f in cands[d]] for d in (0, 1)},
        "primers_after_filters": {str(d): n for d, n in info["candidates"].items()},
        "primers_kept": kept,
        "deleted_by_vertex_cover": sorted(info["deleted"]),
        "csv_sha256": hashlib.sha256(csv.encode()).hexdigest(),
        "report_sha256": hashlib.sha256(report.encode()).hexdigest(),
        "report": report,
        "fasta_sha256": hashlib.sha256(fasta.encode()).hexdigest(),
    }
    Path(a.out).write_text(json.dumps(doc, indent=1) + "\n")
    print("wrote", a.out, "csv lines", csv.count("\n"), flush=True)


if __name__ == "__main__":
    main()
