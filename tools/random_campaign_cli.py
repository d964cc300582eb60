#!/usr/bin/env python3
"""Randomised end-to-end campaign (development aid, GPU): synthetic alignments of random shape through the whole
od-msspe-hip run (stage A -> B -> C -> vertex cover -> CSV + coverage report) with random flag values, against the
oracle-based restatement of main.rs, byte for byte.  usage: random_campaign_cli.py [seed] [cases]"""
import ctypes as C
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "open-msspe-design_amd"))
sys.path.insert(0, str(ROOT / "oracle"))
import numpy as np
import msspe_amd
import ref_pipeline

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 21)
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 25
msspe_amd.load_library()
host = C.CDLL(str(ROOT / "open-msspe-design_amd" / "libod_msspe_host.so"))
bad = 0
for it in range(cases):
    rows = int(rng.integers(3, 90))
    L = int(rng.integers(1500, 9000))
    g = msspe_amd.synth.aligned_genomes(rows, L, seed=int(rng.integers(1, 1 << 30))) \
        if "seed" in msspe_amd.synth.aligned_genomes.__code__.co_varnames else msspe_amd.synth.aligned_genomes(rows, L)
    if it % 2:   # a few gap runs
        for _ in range(int(rng.integers(1, 6))):
            r = int(rng.integers(0, rows)); p0 = int(rng.integers(0, L)); g[r, p0:p0 + int(rng.integers(1, 60))] = ord("-")
    flags, kw = [], {}
    def opt(flag, key, value, text=None):
        flags.extend([flag, text if text is not None else str(value)]); kw[key] = value
    k = int(rng.choice([10, 12, 13, 13, 13, 14, 16, 18]))
    opt("--kmer-size", "kmer_size", k)
    if rng.random() < 0.5: opt("--max-iterations", "max_iterations", int(rng.choice([15, 80, 400])))
    if rng.random() < 0.4: opt("--max-mismatch-segments", "max_mismatch_segments", int(rng.integers(1, 6)))
    if rng.random() < 0.4:
        w = int(rng.choice([300, 400, 700])); opt("--window-size", "window_size", w); opt("--overlap-size", "overlap_size", int(w // 2))
    if rng.random() < 0.3: opt("--search-windows-size", "search_windows_size", int(rng.choice([30, 40, 60])))
    if rng.random() < 0.5: opt("--delta-g-threshold", "dg", float(rng.choice([-3000.0, -5000.0, -7000.0])))
    if rng.random() < 0.3: opt("--annealing-temp", "temp", float(rng.choice([30.0, 37.0, 45.0])))
    if rng.random() < 0.3: opt("--mv-conc", "mv", float(rng.choice([20.0, 100.0])))
    if rng.random() < 0.3: opt("--dv-conc", "dv", float(rng.choice([0.0, 1.5])))
    if rng.random() < 0.3: opt("--tm-stddev", "tm_stddev", float(rng.choice([1.0, 3.0])))
    if rng.random() < 0.3: opt("--max-tm", "max_tm", float(rng.choice([55.0, 70.0])))
    if rng.random() < 0.2: opt("--min-tm", "min_tm", float(rng.choice([20.0, 35.0])))
    if rng.random() < 0.3: opt("--check-hairpin", "check_hairpin", bool(rng.integers(0, 2)), None)
    if flags and flags[-2] == "--check-hairpin": flags[-1] = "true" if kw["check_hairpin"] else "false"
    if rng.random() < 0.2: flags.extend(["--disable-tm-stddev", "true"]); kw["disable_tm_stddev"] = True
    if rng.random() < 0.15: flags.extend(["--keep-all", "true"]); kw["keep_all"] = True
    fasta = "".join(f">g{i} x\n{bytes(r).decode()}\n" for i, r in enumerate(g))
    with tempfile.TemporaryDirectory() as d:
        fa, csv = Path(d) / "in.fa", Path(d) / "out.csv"
        fa.write_text(fasta)
        args = ["od-msspe-hip", "-i", str(fa), "-o", str(csv), "--do-align", "false", *flags]
        arr = (C.c_char_p * len(args))(*[a.encode() for a in args])
        buf = C.create_string_buffer(1 << 22)
        rc = host.odm_run_cli(len(args), arr, buf, 1 << 22)
        got_csv = csv.read_text() if csv.exists() else ""
    try:
        want_csv, want_report, _ = ref_pipeline.run(fasta, **kw)
        ok = rc == 0 and got_csv == want_csv and buf.value.decode() == want_report
        note = f"primers {want_csv.count(chr(10)) - 1}"
    except Exception as e:   # the restatement panics where the reference does: the CLI must fail too
        ok = rc != 0
        note = f"both fail ({type(e).__name__})" if ok else f"restatement raised {e!r}, CLI rc {rc}"
    print(it, f"rows {rows} L {L}", " ".join(flags), "|", note, ok, flush=True)
    bad += not ok
print("BAD", bad)
sys.exit(1 if bad else 0)
