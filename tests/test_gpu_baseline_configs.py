"""BASELINE.json's configurations at their full sizes, on the GPU, through the C ABI / the CLI.

configs[1]  1,000 synthetic aligned 30 kb genomes, Tm + hairpin + self-dimer filters: the whole CLI run
            against the oracle-based restatement of main.rs (CSV + coverage report byte for byte).
configs[2]  10,000 x 30 kb: stage A's first winners re-counted on the CPU, independently of the oracle's
            restatement (which needs minutes at this size): each winner's frequency is the number of live
            segments holding it AND the maximum over all k-mers of the live segments (main.rs:285-329).
headline    65,536-primer pool, 4.29e9 ordered pairs: counts are the popcounts of the bitmap rows, and 32
            sampled rows equal the oracle's decisions.
"""
import ctypes as C
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


@pytest.fixture(scope="module")
def m():
    import msspe_amd
    return msspe_amd


def test_config1_1000_genomes_cli_equals_the_restated_pipeline(m, tmp_path):
    import ref_pipeline
    m.load_library()
    host = C.CDLL(str(ROOT / "open-msspe-design_amd" / "libod_msspe_host.so"))
    g = m.synth.aligned_genomes(1000, 30000)
    fasta = "".join(f">genome{i}\n{bytes(r).decode()}\n" for i, r in enumerate(g))
    fa, csv = tmp_path / "in.fa", tmp_path / "out.csv"
    fa.write_text(fasta)
    args = ["od-msspe-hip", "-i", str(fa), "-o", str(csv), "--do-align", "false", "--check-hairpin", "true",
            "--check-self-dimers", "true"]
    arr = (C.c_char_p * len(args))(*[a.encode() for a in args])
    buf = C.create_string_buffer(1 << 22)
    rc = host.odm_run_cli(len(args), arr, buf, 1 << 22)
    assert rc == 0, buf.value.decode()
    want_csv, want_report, _info = ref_pipeline.run(fasta)
    assert csv.read_text() == want_csv
    assert buf.value.decode() == want_report
    assert want_csv.count("\n") > 100          # hundreds of primers survive the filters


def _head_window_keys(genomes: np.ndarray, seg=500, stride=250, win=50, k=13):
    """2k-bit keys of every k-mer of every head window: (segments, win - k + 1) uint32, 0xffffffff where a
    k-mer covers a non-ACGT column (main.rs:163-171, 173-187)."""
    n, length = genomes.shape
    code = np.full(256, 4, dtype=np.uint8)
    for q, ch in enumerate(b"ACGT"):
        code[ch] = q
    parts = (length - seg) // stride + 1
    starts = np.arange(parts) * stride
    cols = starts[:, None] + np.arange(win)[None, :]                   # (parts, win)
    w = code[genomes[:, cols]]                                          # (n, parts, win)
    w = w.reshape(n * parts, win)
    npos = win - k + 1
    keys = np.zeros((w.shape[0], npos), dtype=np.uint32)
    bad = np.zeros((w.shape[0], npos), dtype=bool)
    for q in range(k):
        col = w[:, q:q + npos]
        keys = (keys << np.uint32(2)) | (col & 3).astype(np.uint32)
        bad |= col > 3
    keys[bad] = 0xFFFFFFFF
    return keys, parts


def test_config2_10000_genomes_stage_a_winners_are_maximal_and_counted_right(m):
    n_rows, length, k, check = 10000, 30000, 13, 24
    genomes = m.synth.aligned_genomes(n_rows, length)
    eng = m.Engine(0)
    try:
        words, freqs = eng.kmer_candidates(genomes, m.KmerOpt(500, 250, 50, k, 1000, 10), 0)
    finally:
        eng.close()
    freqs = np.asarray(freqs, dtype=np.int64)
    assert len(words) > 300 and len(set(words)) == len(words)
    assert np.all(np.diff(freqs) <= 0)                    # a greedy max-cover never finds a larger set later
    assert freqs[-1] >= 1
    keys, parts = _head_window_keys(genomes)
    # duplicates inside a window count once (itertools::unique, main.rs:168): sort each row, blank repeats
    keys.sort(axis=1)
    keys[:, 1:][keys[:, 1:] == keys[:, :-1]] = 0xFFFFFFFF
    live = np.ones(keys.shape[0], dtype=bool)
    code = {c: q for q, c in enumerate("ACGT")}
    for t in range(check):
        wkey = 0
        for ch in words[t]:
            wkey = (wkey << 2) | code[ch]
        holds = (keys == np.uint32(wkey)).any(axis=1)
        assert int((holds & live).sum()) == freqs[t], (t, words[t])
        flat = keys[live].ravel()
        flat = flat[flat != 0xFFFFFFFF]
        top = np.bincount(flat, minlength=1 << (2 * k)).max()
        assert top == freqs[t], (t, words[t], top)         # no k-mer of the live segments is more frequent
        live &= ~holds                                      # main.rs:371-378: every holder is ignored from now on


def test_headline_pool_65536_counts_bitmap_and_sampled_rows(m, oracle, oracle_tables):
    import torch
    n, k = 65536, 13
    pool_ascii = m.synth.random_pool(n, k)
    d_pool = torch.from_numpy(m.pack_oligos(pool_ascii).view(np.int64)).cuda()
    d_rc = torch.zeros(n, dtype=torch.int32, device="cuda")
    d_bm = torch.zeros((n, n // 64), dtype=torch.int64, device="cuda")
    eng = m.Engine(0)
    try:
        eng.set_stream(torch.cuda.current_stream().cuda_stream)
        eng.cross_dimer_dev(d_pool.data_ptr(), n, k, m.Chem.ntthal(), -9000.0, (0, n), (0, n),
                            d_rc.data_ptr(), d_bm.data_ptr())
        torch.cuda.synchronize()
        stats = eng.pair_stage_stats()
    finally:
        eng.close()
    rc = d_rc.cpu().numpy().astype(np.int64)
    bm = d_bm.cpu().numpy().view(np.uint64)
    np.testing.assert_array_equal(np.bitwise_count(bm).sum(axis=1).astype(np.int64), rc)
    assert 0.004 < rc.sum() / float(n) ** 2 < 0.007          # 0.54 % of random 13-mer pairs conflict
    assert stats["replay_mismatch"] == 0 and stats["list"]["replay_mismatch"] == 0
    rows = np.random.default_rng(65536).choice(n, 32, replace=False)
    for r in rows:
        _, _, cf, _ = oracle.pool_pairs(oracle_tables, pool_ascii, rows=(int(r), int(r) + 1), want_dg=False)
        got = np.unpackbits(bm[r].view(np.uint8), bitorder="little")[:n]
        np.testing.assert_array_equal(got, cf[0])
