#!/usr/bin/env python3
"""End-to-end run of the od-msspe pipeline (BASELINE.json configs[0]/[1] shapes) through the host
layer on the GPU, timed next to the oracle-based CPU restatement; checks the CSVs are identical."""
import ctypes as C
import sys
import tempfile
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "open-msspe-design_amd"))
sys.path.insert(0, str(ROOT / "oracle"))
import msspe_amd
import ref_pipeline

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 10
length = int(sys.argv[2]) if len(sys.argv) > 2 else 29903
do_cpu = (sys.argv[3] != "nocpu") if len(sys.argv) > 3 else True
msspe_amd.load_library()
host = C.CDLL(str(ROOT / "open-msspe-design_amd" / "libod_msspe_host.so"))
g = msspe_amd.synth.aligned_genomes(rows, length)
fasta = "".join(f">genome{i}\n{bytes(r).decode()}\n" for i, r in enumerate(g))
with tempfile.TemporaryDirectory() as d:
    fa, csv = Path(d) / "in.fa", Path(d) / "out.csv"
    fa.write_text(fasta)
    args = ["od-msspe-hip", "-i", str(fa), "-o", str(csv), "--do-align", "false"]
    arr = (C.c_char_p * len(args))(*[a.encode() for a in args])
    buf = C.create_string_buffer(1 << 20)
    for rep in range(2):
        t0 = time.time()
        rc = host.odm_run_cli(len(args), arr, buf, 1 << 20)
        dt_gpu = time.time() - t0
    assert rc == 0, buf.value.decode()
    gpu_csv = csv.read_text()
    print(f"{rows} genomes x {length}: GPU pipeline {dt_gpu*1e3:.0f} ms, {gpu_csv.count(chr(10)) - 1} primers", flush=True)
    print(buf.value.decode())
    if do_cpu:
        t0 = time.time()
        want_csv, want_report, info = ref_pipeline.run(fasta)
        dt_cpu = time.time() - t0
        print(f"CPU restatement (oracle, mostly 1 thread): {dt_cpu:.1f} s; CSV identical: {want_csv == gpu_csv}; "
              f"report identical: {want_report == buf.value.decode()}")
