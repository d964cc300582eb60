"""Stage A on the GPU (msspe_kmer_candidates) vs. the oracle's restatement of
od-msspe/src/main.rs:196-406: winners and frequencies must be identical, in order."""
import json

import numpy as np
import pytest

from helpers import window_with_kmers

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    import msspe_amd
    e = msspe_amd.Engine(0)
    yield e
    e.close()


@pytest.fixture(scope="module")
def m():
    import msspe_amd
    return msspe_amd


def run_both(eng, m, oracle, seqs, seg=500, stride=250, win=50, k=13, iters=1000, mm=1):
    arr = np.frombuffer("".join(seqs).encode(), dtype=np.uint8).reshape(len(seqs), -1)
    segs = oracle.Segments(seqs, seg, stride, win, k)
    for d in (0, 1):
        opt = m.KmerOpt(seg, stride, win, k, iters, mm)
        words, freqs = eng.kmer_candidates(arr, opt, d)
        want = segs.candidates(d, iters, mm)
        assert list(zip(words, freqs.tolist())) == want, f"direction {d}"
    return segs


def test_reference_unit_vectors_through_the_c_abi(eng, m, oracle, golden_dir):
    """main.rs:1138-1235 (winner ACT, frequency 2) with the hand-built segments."""
    g = json.loads((golden_dir / "stage_a_unit.json").read_text())["most_freq"]
    genome = ""
    for s in g["segments"]:
        tail = [oracle.reverse_complement(w) for w in reversed(s["rev"])]
        genome += window_with_kmers(s["fwd"], 11) + window_with_kmers(tail, 11)
    arr = np.frombuffer(genome.encode(), dtype=np.uint8).reshape(1, -1)
    words, freqs = eng.kmer_candidates(arr, m.KmerOpt(22, 22, 11, 3, 1, 1), 0)
    assert (words, freqs.tolist()) == ([g["winner"]], [g["frequency"]])


@pytest.mark.parametrize("n_rows,length,mm", [(10, 3000, 1), (40, 6000, 1), (120, 4000, 3)])
def test_synthetic_alignments(eng, m, oracle, n_rows, length, mm):
    genomes = m.synth.aligned_genomes(n_rows, length)
    seqs = [bytes(r).decode() for r in genomes]
    segs = run_both(eng, m, oracle, seqs, mm=mm)
    assert len(segs) == n_rows * ((length - 500) // 250 + 1)


def test_small_parameters_and_ties(eng, m, oracle):
    """Short segments and k = 3 produce many frequency ties: exercises partition_tie_score and
    the lexicographic tie-break, plus the max_iterations cap."""
    rng = np.random.default_rng(5)
    anc = rng.integers(0, 4, 400)
    seqs = []
    for _ in range(25):
        row = anc.copy()
        mut = rng.random(400) < 0.05
        row[mut] = rng.integers(0, 4, int(mut.sum()))
        s = "".join("ACGT"[x] for x in row)
        seqs.append(s[:120] + "-" * 7 + s[127:300] + "N" * 3 + s[303:])
    run_both(eng, m, oracle, seqs, seg=40, stride=20, win=12, k=3, iters=1000, mm=1)
    run_both(eng, m, oracle, seqs, seg=40, stride=20, win=12, k=5, iters=7, mm=1)
    run_both(eng, m, oracle, seqs, seg=60, stride=30, win=30, k=8, iters=1000, mm=4)


@pytest.mark.parametrize("seed", range(8))
def test_mixed_tie_rates(eng, m, oracle, seed):
    """Alignments whose rate of frequency ties ranges from rare to constant (single candidates skip
    the scoring, ties go through partition_tie_score), with and without a cap on the iterations
    (the loop ends in mid-batch of its graph replays)."""
    rng = np.random.default_rng(100 + seed)
    rows, length = int(rng.integers(12, 90)), int(rng.integers(1500, 5000))
    rate = [0.002, 0.01, 0.03, 0.08, 0.15, 0.3, 0.01, 0.05][seed]
    anc = rng.integers(0, 4, length)
    seqs = []
    for _ in range(rows):
        row = anc.copy()
        mut = rng.random(length) < rate
        row[mut] = rng.integers(0, 4, int(mut.sum()))
        seqs.append("".join("ACGT"[x] for x in row))
    k = int(rng.choice([6, 9, 13]))
    run_both(eng, m, oracle, seqs, seg=200, stride=100, win=40, k=k, iters=1000, mm=int(rng.integers(1, 4)))
    run_both(eng, m, oracle, seqs, seg=200, stride=100, win=40, k=k, iters=int(rng.integers(1, 70)), mm=1)


def test_rows_uploaded_through_pinned_staging(eng, m, oracle):
    """msspe_device_put_rows: ragged rows (one string per record), padded with '-' on the way to the
    device, enough of them for several staging chunks; stage A on that buffer equals stage A on the
    rectangular host array and the oracle."""
    import ctypes as C
    rng = np.random.default_rng(11)
    base = m.synth.aligned_genomes(700, 30000)
    rows = [bytes(r[: 30000 - int(rng.integers(0, 400))]) for r in base]     # 21 MB, ragged
    rows[3] = b""
    L = 30000
    arr = np.full((len(rows), L), ord("-"), dtype=np.uint8)
    for i, r in enumerate(rows):
        arr[i, : len(r)] = np.frombuffer(r, dtype=np.uint8)
    ptrs = (C.c_char_p * len(rows))(*rows)
    lens = (C.c_size_t * len(rows))(*[len(r) for r in rows])
    dev = C.c_void_p()
    eng.L.msspe_device_put_rows.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.c_int,
                                            C.c_size_t, C.c_int, C.POINTER(C.c_void_p)]
    eng.L.msspe_device_free.argtypes = [C.c_void_p, C.c_void_p]
    assert eng.L.msspe_device_put_rows(eng.ptr, ptrs, lens, len(rows), L, ord("-"), C.byref(dev)) == 0
    try:
        opt = m.KmerOpt(500, 250, 50, 13, 40, 3)
        for d in (0, 1):
            got = eng.kmer_candidates(None, opt, d, device_ptr=dev.value, n_seq=len(rows), seq_len=L)
            want = eng.kmer_candidates(arr, opt, d)
            assert got[0] == want[0] and got[1].tolist() == want[1].tolist() and len(got[0]) == 40
    finally:
        assert eng.L.msspe_device_free(eng.ptr, dev) == 0
    seqs = [bytes(r).decode() for r in arr[:60]]
    run_both(eng, m, oracle, seqs, iters=15, mm=2)


def test_plain_launches_instead_of_graph_replays(eng, m, oracle, monkeypatch):
    """The greedy loop's fallback when hipGraph capture is not available."""
    eng.set_option("stage_a_graph", 0)
    try:
        seqs = [bytes(r).decode() for r in m.synth.aligned_genomes(30, 4000)]
        run_both(eng, m, oracle, seqs, mm=2)
    finally:
        eng.set_option("stage_a_graph", 1)


@pytest.mark.parametrize("k", [15, 16, 22, 31])
def test_long_words_take_the_64_bit_sort_keys(eng, m, oracle, k):
    """Words of up to 15 bases sort as 32-bit keys (2k + 1 bits), longer ones as 64-bit keys: both sides of
    the switch and the longest word the packed form holds, against the oracle."""
    seqs = [bytes(r).decode() for r in m.synth.aligned_genomes(60, 5000)]
    run_both(eng, m, oracle, seqs, seg=500, stride=250, win=max(50, k + 20), k=k, iters=200, mm=2)


def test_many_partitions_take_the_all_words_loop(eng, m, oracle):
    """More partitions than the candidate-list kernel keeps bitmaps for (8,192): the loop runs the all-words
    batches; same winners as the oracle."""
    rng = np.random.default_rng(3)
    anc = rng.integers(0, 4, 172000)
    seqs = []
    for r in range(3):
        row = anc.copy()
        mut = rng.random(anc.size) < 0.02
        row[mut] = rng.integers(0, 4, int(mut.sum()))
        seqs.append("".join("ACGT"[x] for x in row))
    run_both(eng, m, oracle, seqs, seg=40, stride=20, win=20, k=6, iters=120, mm=1)   # 8,599 partitions


def test_candidate_list_loop_equals_the_all_words_loop(eng, m, oracle):
    """The two greedy-loop drivers (option stage_a_candidates: the list of words near the maximum, rebuilt
    whenever the maximum halves, against a scan of every word on every iteration) pick the same winners
    with the same frequencies, to the last iteration (frequency 2), with graph replays and without."""
    cases = [(m.synth.aligned_genomes(400, 9000), m.KmerOpt(500, 250, 50, 13, 1000, 1)),
             (m.synth.aligned_genomes(120, 3000), m.KmerOpt(300, 100, 40, 6, 1000, 1)),
             (m.synth.aligned_genomes(900, 30000), m.KmerOpt(500, 250, 50, 13, 300, 3))]
    for arr, opt in cases:
        for d in (0, 1):
            res = {}
            for cand, graph in ((0, 1), (1, 1), (1, 0)):
                eng.set_option("stage_a_candidates", cand)
                eng.set_option("stage_a_graph", graph)
                try:
                    got = eng.kmer_candidates(arr, opt, d)
                finally:
                    eng.set_option("stage_a_candidates", 1)
                    eng.set_option("stage_a_graph", 1)
                res[(cand, graph)] = (list(got[0]), got[1].tolist())
            assert res[(0, 1)] == res[(1, 1)] == res[(1, 0)]
            assert len(res[(1, 1)][0]) > 20


def test_candidate_list_too_long_falls_back(eng, m, oracle):
    """Unrelated genomes, each present twice: a quarter of a million words tie at frequency 2, far more than
    the candidate-list kernel reads per block, so the loop runs the all-words batches (and retries the list
    every few batches).  Same winners as the all-words loop alone, and as the oracle on a small case."""
    rng = np.random.default_rng(23)
    half = rng.integers(0, 4, (300, 6000))
    arr = np.frombuffer(b"ACGT", dtype=np.uint8)[np.concatenate([half, half])]
    opt = m.KmerOpt(500, 250, 50, 13, 900, 1)   # a few hundred chance repeats (frequency 4, 6) come first
    res = []
    for cand in (0, 1):
        eng.set_option("stage_a_candidates", cand)
        try:
            got = eng.kmer_candidates(arr, opt, 0)
        finally:
            eng.set_option("stage_a_candidates", 1)
        res.append((list(got[0]), got[1].tolist()))
    assert res[0] == res[1] and len(res[0][0]) == 900 and res[0][1][-200:] == [2] * 200
    small = np.frombuffer(b"ACGT", dtype=np.uint8)[np.concatenate([half[:6, :2000], half[:6, :2000]])]
    run_both(eng, m, oracle, [bytes(r).decode() for r in small], iters=60, mm=1)


def test_long_posting_lists_and_ties(eng, m, oracle):
    """700 near-identical rows: posting lists of thousands of segments spread over many partitions
    (k = 3) tie at the same frequency, so the block-per-word scoring kernel, its ordered path and
    the wave-merged count updates all decide winners here."""
    rng = np.random.default_rng(11)
    anc = rng.integers(0, 4, 400)
    seqs = []
    for r in range(700):
        row = anc.copy()
        mut = rng.random(400) < (0.002 if r % 3 else 0.03)
        row[mut] = rng.integers(0, 4, int(mut.sum()))
        seqs.append("".join("ACGT"[x] for x in row))
    run_both(eng, m, oracle, seqs, seg=40, stride=20, win=12, k=3, iters=1000, mm=1)
    run_both(eng, m, oracle, seqs, seg=60, stride=30, win=30, k=8, iters=1000, mm=20)
    run_both(eng, m, oracle, seqs, seg=100, stride=50, win=50, k=13, iters=50, mm=1)


def test_output_capacity(eng, m, oracle):
    """The loop state lives on the device; a caller buffer that is too small is an error, one that
    is exactly large enough is not."""
    genomes = m.synth.aligned_genomes(40, 6000)
    opt = m.KmerOpt(500, 250, 50, 13, 1000, 1)
    words, freqs = eng.kmer_candidates(genomes, opt, 0)
    assert len(words) > 8
    w2, f2 = eng.kmer_candidates(genomes, opt, 0, capacity=len(words))
    assert (w2, f2.tolist()) == (words, freqs.tolist())
    with pytest.raises(m.MsspeError):
        eng.kmer_candidates(genomes, opt, 0, capacity=len(words) - 1)
    w3, f3 = eng.kmer_candidates(genomes, m.KmerOpt(500, 250, 50, 13, 5, 1), 0, capacity=5)
    assert (w3, f3.tolist()) == (words[:5], freqs[:5].tolist())


def test_edge_inputs(eng, m, oracle):
    opt = m.KmerOpt(500, 250, 50, 13, 1000, 1)
    short = np.frombuffer(("ACGT" * 100).encode(), dtype=np.uint8).reshape(1, -1)   # < one segment
    assert eng.kmer_candidates(short, opt, 0)[0] == []
    gaps = np.full((4, 1000), ord("-"), dtype=np.uint8)                             # no valid k-mer
    assert eng.kmer_candidates(gaps, opt, 1)[0] == []
    with pytest.raises(m.MsspeError):                                               # stride < window
        eng.kmer_candidates(gaps, m.KmerOpt(500, 40, 50, 13, 10, 1), 0)


def test_segment_coverage_matches_a_string_search(eng, m, oracle):
    """msspe_segment_coverage vs. the reference's rule (main.rs:518-594): head window holds a
    forward primer, or the tail window holds the reverse complement of a reverse primer."""
    genomes = m.synth.aligned_genomes(30, 5000)
    seqs = [bytes(r).decode() for r in genomes]
    segs = oracle.Segments(seqs, 500, 250, 50, 13)
    fwd = [w for w, _ in segs.candidates(0, 12, 1)]
    rev = [w for w, _ in segs.candidates(1, 9, 1)]
    opt = m.KmerOpt(500, 250, 50, 13, 0, 0)
    for f, r in ((fwd, rev), (fwd, []), ([], rev), ([], [])):
        hit = eng.segment_coverage(genomes, opt, f, r)
        want = np.zeros_like(hit)
        sf, sr = set(f), set(r)
        for i, s in enumerate(seqs):
            for j in range(hit.shape[1]):
                part = s[j * 250: j * 250 + 500]
                head = oracle.find_kmers(part[:50], 13)
                tail = oracle.find_kmers(part[-50:], 13)
                want[i, j] = any(w in sf for w in head) or any(oracle.reverse_complement(w) in sr for w in tail)
        np.testing.assert_array_equal(hit, want)
    assert hit.shape == (30, (5000 - 500) // 250 + 1)


def test_packed_alignment_equals_the_ascii_path(eng, m, oracle):
    """msspe_device_put_rows_packed (2-bit bases + validity bit, packed on the device) and the *_packed_dev
    entry points: the packed words are what a host-side packing gives, stage A's winners and the coverage
    hits are identical to the byte-matrix path (ragged rows, gaps, N, IUPAC codes, several upload chunks)."""
    import ctypes as C
    import torch
    rng = np.random.default_rng(5)
    base = m.synth.aligned_genomes(650, 30000)
    base[7, 1000:1040] = np.frombuffer(b"RYKMSWBDHVRYKMSWBDHVRYKMSWBDHVRYKMSWBDHV", dtype=np.uint8)
    rows = [bytes(r[: 30000 - int(rng.integers(0, 300))]) for r in base]
    L = 30000
    arr = np.full((len(rows), L), ord("-"), dtype=np.uint8)
    for i, r in enumerate(rows):
        arr[i, : len(r)] = np.frombuffer(r, dtype=np.uint8)
    ptrs = (C.c_char_p * len(rows))(*rows)
    lens = (C.c_size_t * len(rows))(*[len(r) for r in rows])
    dev = C.c_void_p()
    eng.L.msspe_device_free.argtypes = [C.c_void_p, C.c_void_p]
    assert eng.L.msspe_device_put_rows_packed(eng.ptr, ptrs, lens, len(rows), L, C.byref(dev)) == 0
    try:
        rw = int(eng.L.msspe_packed_row_words(L))
        bw = (L + 31) // 32
        assert rw == bw + (L + 63) // 64
        # the packed image against numpy
        code = np.full(256, 4, dtype=np.uint8)
        for q, ch in enumerate(b"ACGT"):
            code[ch] = q
        c = code[arr]
        pad = np.zeros((len(rows), bw * 32 - L), dtype=np.uint8)
        bases = (np.concatenate([c & 3, pad], 1).reshape(len(rows), bw, 32).astype(np.uint64)
                 << (2 * np.arange(32, dtype=np.uint64))[None, None, :]).sum(2, dtype=np.uint64)
        vpad = np.zeros((len(rows), (rw - bw) * 64 - L), dtype=np.uint8)
        valid = (np.concatenate([(c < 4).astype(np.uint8), vpad], 1).reshape(len(rows), rw - bw, 64).astype(np.uint64)
                 << np.arange(64, dtype=np.uint64)[None, None, :]).sum(2, dtype=np.uint64)
        got_words = np.empty((len(rows), rw), dtype=np.uint64)
        torch.cuda.synchronize()
        import ctypes
        hip = ctypes.CDLL("libamdhip64.so")
        assert hip.hipMemcpy(ctypes.c_void_p(got_words.ctypes.data), dev, ctypes.c_size_t(got_words.nbytes), 2) == 0
        np.testing.assert_array_equal(got_words[:, :bw], bases)
        np.testing.assert_array_equal(got_words[:, bw:], valid)
        # stage A and the coverage report's segment search on the packed form
        opt = m.KmerOpt(500, 250, 50, 13, 60, 3)
        words = np.zeros(60, dtype=np.uint64)
        freqs = np.zeros(60, dtype=np.uint32)
        winners = {}
        for d in (0, 1):
            n_out = C.c_int(0)
            assert eng.L.msspe_kmer_candidates_packed_dev(eng.ptr, dev, len(rows), L, C.byref(opt), d, words.ctypes.data,
                                                          freqs.ctypes.data, 60, C.byref(n_out)) == 0
            want = eng.kmer_candidates(arr, opt, d)
            got = [m.unpack_oligo(w, 13) for w in words[: n_out.value]]
            assert got == want[0] and freqs[: n_out.value].tolist() == want[1].tolist() and len(got) == 60
            winners[d] = words[: n_out.value].copy()
        P = (L - 500) // 250 + 1
        hit_p = np.zeros(len(rows) * P, dtype=np.uint8)
        hit_a = np.zeros(len(rows) * P, dtype=np.uint8)
        assert eng.L.msspe_segment_coverage_packed_dev(eng.ptr, dev, len(rows), L, C.byref(opt), winners[0].ctypes.data, 60,
                                                       winners[1].ctypes.data, 60, hit_p.ctypes.data) == 0
        assert eng.L.msspe_segment_coverage(eng.ptr, arr.ctypes.data, len(rows), L, C.byref(opt), winners[0].ctypes.data, 60,
                                            winners[1].ctypes.data, 60, hit_a.ctypes.data) == 0
        np.testing.assert_array_equal(hit_p, hit_a)
        assert 0 < hit_p.sum() < hit_p.size
    finally:
        assert eng.L.msspe_device_free(eng.ptr, dev) == 0


def test_the_loop_graph_is_kept_and_replaced_as_the_arguments_change(eng, m, oracle):
    """The candidate-list loop's graph outlives a call (kmer_stage.hpp: loop_exec_) and serves the next one if every
    kernel argument is unchanged -- the two directions of one alignment, a repeated call -- and is captured anew when
    the shape, the window or the winner capacity changes.  Shapes alternate here so that every call either reuses
    the graph of the same shape's earlier call or replaces another shape's."""
    shapes = [(60, 5000, dict()), (90, 3500, dict(win=40)), (60, 5000, dict()), (60, 5000, dict(iters=300)),
              (90, 3500, dict(win=40)), (25, 5000, dict(k=11))]
    seen = {}
    for n_rows, length, kw in shapes:
        key = (n_rows, length, tuple(sorted(kw.items())))
        if key not in seen:
            seen[key] = [bytes(r).decode() for r in m.synth.aligned_genomes(n_rows, length, seed=len(seen) + 3)]
        run_both(eng, m, oracle, seen[key], **kw)
