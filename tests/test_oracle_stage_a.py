"""Oracle vs. the reference's stage-A unit tests (od-msspe/src/main.rs:868-1235); no GPU."""
import json

import pytest

from helpers import window_with_kmers


@pytest.fixture(scope="module")
def g(golden_dir):
    return json.loads((golden_dir / "stage_a_unit.json").read_text())


def test_reverse_complement(oracle, g):
    v = g["reverse_complement"]
    assert oracle.reverse_complement(v["input"]) == v["expected"]
    assert oracle.reverse_complement("AUN-") == "-NAT"   # U -> A, others unchanged (main.rs:152-159)


def test_search_windows_and_find_kmers(oracle, g):
    v = g["search_windows"]
    s, w = v["sequence"], v["window"]
    assert (s[:w], s[len(s) - w:]) == (v["first"], v["second"])
    v = g["find_kmers"]
    assert oracle.find_kmers(v["sequence"], v["k"]) == v["expected"]
    assert oracle.find_kmers("ACGTACGTAC", 4) == ["ACGT", "CGTA", "GTAC", "TACG"]   # dedup keeps first
    assert oracle.find_kmers("ACG", 4) == []


def test_partitioning(oracle, g):
    v = g["partitioning"]
    assert oracle.partitions(v["sequence"], v["size"], v["stride"]) == v["expected"]
    assert oracle.partitions("ACGT", 10, 5) == []


def test_get_segments(oracle, g):
    v = g["get_segments"]
    m = oracle.Segments([s for _, s in v["records"]], v["segment_size"], v["overlap_size"],
                        v["window_size"], v["kmer_size"])
    assert len(m) == v["n_segments"]
    assert len(m.kmers(0, 0)) == v["segment0_fwd_kmers"]
    assert len(m.kmers(1, 1)) == v["segment1_rev_kmers"]
    assert [m.partition_no(s) for s in range(6)] == [0, 1, 0, 1, 0, 1]
    # search windows = (AACCT)(TGGAA): fwd AAC ACC CCT ; rev-comp of TGG GGA GAA = CCA TCC TTC
    assert m.kmers(0, 0) == ["AAC", "ACC", "CCT"]
    assert m.kmers(0, 1) == ["CCA", "TCC", "TTC"]
    with pytest.raises(ValueError):
        oracle.Segments(["A" * 20], 10, 4, 5, 3)     # stride < window panics (main.rs:201-203)


def _hand_built(oracle, segs, k=3, width=11):
    """One genome whose partitions reproduce hand-built segments (head = fwd k-mers, tail =
    reverse complement of the listed rev k-mers, reversed back into window order)."""
    genome = ""
    for s in segs:
        tail_words = [oracle.reverse_complement(w) for w in reversed(s["rev"])]
        genome += window_with_kmers(s["fwd"], width) + window_with_kmers(tail_words, width)[::-1][::-1]
    return oracle.Segments([genome], 2 * width, 2 * width, width, k)


def test_make_kmer_segments_mapping(oracle, g):
    v = g["mapping"]
    m = _hand_built(oracle, v["segments"])
    assert len(m) == 2
    for si, s in enumerate(v["segments"]):
        assert sorted(m.kmers(si, 0)) == sorted(s["fwd"])
        assert sorted(m.kmers(si, 1)) == sorted(s["rev"])
    assert m.mapping_key_count() == v["n_keys"]
    for key, n in v["postings"].items():
        word, d = key.split("/")
        assert m.postings(word, int(d)) == n


def test_find_most_freq_kmer(oracle, g):
    v = g["most_freq"]
    m = _hand_built(oracle, v["segments"])
    assert [m.partition_no(s) for s in range(2)] == [s["partition_no"] for s in v["segments"]]
    assert m.most_freq(v["direction"]) == (v["winner"], v["frequency"])


def test_find_candidates_semantics(oracle):
    """SURVEY.md A.5 (no reference test covers find_candidates_kmers as a whole): stop rules."""
    # three identical genomes, one partition each: every head k-mer has frequency 3
    seq = "ACGTTGCAAGGCTTAACCGGATCGATCGGATTACAGGCTTAACGTACGATCGTAGCTAGCTAGGATCCGAT"[:60]
    m = oracle.Segments([seq] * 3, 60, 60, 20, 5)
    win = m.candidates(0, max_iterations=10, max_mismatch_segments=1)
    # first winner covers all three segments; afterwards nothing is left -> loop stops (None)
    assert win == [(min(m.kmers(0, 0)), 3)]
    # frequency 1 stops before the push
    other = "TTTTTGGGGGCCCCCAAAAA" + "T" * 40
    m = oracle.Segments([seq, other], 60, 60, 20, 5)
    assert m.candidates(0, 10, 1) == []
    # frequency < max_mismatch_segments stops after the push
    m = oracle.Segments([seq] * 3, 60, 60, 20, 5)
    assert len(m.candidates(0, 10, 5)) == 1


def test_is_run_only_looks_at_the_trailing_run(oracle):
    """main.rs:478-490: counter resets on every change and is tested once after the loop."""
    assert oracle.is_run("ACGTAAAAAA")          # 6 trailing A
    assert not oracle.is_run("AAAAAAACGT")      # leading run is forgotten
    assert not oracle.is_run("ACGTTTTTT"[:8])   # 4 trailing T -> runs = 3
    assert oracle.is_run("CCCCCC") and not oracle.is_run("CCCCC")
