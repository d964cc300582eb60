"""The C++ host layer (open-msspe-design_amd/host/): CLI parsing with od-msspe's flag and env
names, FASTA normalisation, vertex cover, CSV and coverage report -- no GPU needed for these."""
import ctypes as C
import os
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
LIB = ROOT / "open-msspe-design_amd" / "libod_msspe_host.so"


@pytest.fixture(scope="module")
def host():
    import msspe_amd
    msspe_amd.load_library()              # libod_msspe_host.so depends on libmsspe_hip.so
    return C.CDLL(str(LIB))


def call(fn, *args, cap=1 << 20):
    buf = C.create_string_buffer(cap)
    rc = fn(*args, buf, cap)
    return rc, buf.value.decode()


def argv(*a):
    arr = (C.c_char_p * (len(a) + 1))(b"od-msspe-hip", *[x.encode() for x in a])
    return len(a) + 1, arr


def test_cli_flags_follow_config_rs(host, monkeypatch):
    rc, out = call(host.odm_parse_args, *argv("-i", "in.fa", "-o", "out.csv", "--kmer-size", "15",
                                              "--check-hairpin=false", "--delta-g-threshold", "-8000"))
    assert rc == 0
    kv = dict(l.split("=", 1) for l in out.splitlines())
    assert kv["kmer_size"] == "15" and kv["check_hairpin"] == "false" and kv["delta_g_threshold"] == "-8000"
    assert kv["window_size"] == "500" and kv["overlap_size"] == "250" and kv["search_windows_size"] == "50"
    assert kv["max_iterations"] == "1000" and kv["max_mismatch_segments"] == "-1" and kv["do_align"] == "true"
    # a bare boolean flag is a usage error: booleans are "true"/"false" strings (config.rs:65-140)
    rc, out = call(host.odm_parse_args, *argv("-i", "a", "-o", "b", "--check-hairpin"))
    assert rc == 2 and "a value is required" in out
    rc, out = call(host.odm_parse_args, *argv("-i", "a", "-o", "b", "--keep-all", "yes"))
    assert rc == 2 and "possible values: true, false" in out
    rc, out = call(host.odm_parse_args, *argv("-o", "b"))
    assert rc == 2 and "--input <INPUT>" in out
    # every option except -i/-o has an environment fallback; the command line wins
    monkeypatch.setenv("KMER_SIZE", "11")
    monkeypatch.setenv("KEEP_ALL", "true")
    rc, out = call(host.odm_parse_args, *argv("-i", "a", "-o", "b"))
    kv = dict(l.split("=", 1) for l in out.splitlines())
    assert kv["kmer_size"] == "11" and kv["keep_all"] == "true"
    rc, out = call(host.odm_parse_args, *argv("-i", "a", "-o", "b", "--kmer-size", "13"))
    assert dict(l.split("=", 1) for l in out.splitlines())["kmer_size"] == "13"


def test_fasta_normalisation(host):
    """main.rs:108-122: id up to the first blank, lines joined, upper-cased, U -> T."""
    rc, out = call(host.odm_to_records, b">seq1 some description\nacgu\nNN-a\r\n>seq2\nUUUU\n")
    assert out == "seq1\tACGTNN-A\nseq2\tTTTT\n"


def test_fasta_large_input_is_parsed_in_pieces(host):
    """Inputs above 16 MB are cut at header lines and parsed on several host threads: same records,
    same order, whatever the line structure (wrapped lines, CRLF, lower case, U, a header-less head)."""
    rng = np.random.default_rng(5)
    alphabet = np.frombuffer(b"ACGTacgtuUN-", dtype=np.uint8)
    parts, want = ["stray line before any header\n"], []
    for r in range(900):
        seq = alphabet[rng.integers(0, len(alphabet), 20000 + int(rng.integers(0, 9000)))].tobytes().decode()
        width = int(rng.integers(50, 200))
        eol = "\r\n" if r % 7 == 0 else "\n"
        parts.append(f">rec{r} description {r}{eol}" + eol.join(seq[i:i + width] for i in range(0, len(seq), width)) + eol)
        want.append(f"rec{r}\t" + seq.upper().replace("U", "T"))
    fasta = "".join(parts).encode()
    assert len(fasta) > (16 << 20)
    rc, out = call(host.odm_to_records, fasta, cap=len(fasta) + (1 << 20))
    got = out.splitlines()
    assert len(got) == 900 and got == want


def test_vertex_cover_rules(host):
    """main.rs:754-798: most live conflicts first, ties to the lexicographically greatest primer,
    a self-conflicting primer is always removed."""
    import ref_pipeline
    cases = [
        (["AAA", "CCC", "GGG"], [("AAA", "CCC"), ("GGG", "CCC")]),            # hub CCC goes first
        (["AAA", "CCC"], [("AAA", "CCC")]),                                    # tie 1-1: CCC (greater)
        (["AAA", "CCC", "TTT"], [("TTT", "TTT")]),                             # self loop only
        (["AAA", "CCC", "GGG", "TTT"], [("AAA", "CCC"), ("CCC", "AAA"), ("GGG", "TTT"), ("AAA", "AAA")]),
        (["ACG", "ACG", "TTT"], [("ACG", "TTT")]),                             # duplicate words collapse
        (["AAA", "CCC"], []),
    ]
    rng = np.random.default_rng(1)
    words = ["".join("ACGT"[x] for x in rng.integers(0, 4, 5)) for _ in range(14)]
    cases.append((words, [(words[a], words[b]) for a, b in rng.integers(0, 14, (30, 2))]))
    for primers, edges in cases:
        rc, out = call(host.odm_vertex_cover, "\n".join(primers).encode(),
                       "\n".join(f"{a},{b}" for a, b in edges).encode())
        assert rc >= 0
        assert set(out.split()) == ref_pipeline.vertex_cover(primers, set(edges)), (primers, edges)


def test_is_run_and_tm_stat(host, oracle):
    for w in ["ACGTAAAAAA", "AAAAAAACGT", "CCCCCC", "CCCCC", "ACGTTTTTTG"]:
        assert bool(host.odm_is_run(w.encode())) == oracle.is_run(w)
    tm = np.array([43.727, 41.5, 47.25, 39.0, 52.125], dtype=np.float32)
    host.odm_tm_stat.restype = C.c_float
    sd = C.c_float()
    mean = host.odm_tm_stat(tm.ctypes.data_as(C.POINTER(C.c_float)), len(tm), 0, C.byref(sd))
    m2, s2 = oracle.tm_stat(tm, True)
    assert (np.float32(mean), np.float32(sd.value)) == (np.float32(m2), np.float32(s2))


def test_csv_and_coverage_report_text(host, oracle):
    """main.rs:834-858 (two decimals, gc/100, index restarts per direction) and :518-594."""
    import ref_pipeline
    rows = "AGCCCGTGTAAAC,0,53.846,43.25,2.125,43.727\nGGGCCGTGTAAAC,0,61.538,43.25,2.125,47.0\n" \
           "TTTCCGTGTAAAC,1,38.462,40.0,1.0,39.995"
    rc, out = call(host.odm_primers_csv, rows.encode())
    assert out == ("direction,name,primers,gc,avg,std,tm\n"
                   "F,Primer_0_F,AGCCCGTGTAAAC,0.54,43.25,2.12,43.73\n"
                   "F,Primer_1_F,GGGCCGTGTAAAC,0.62,43.25,2.12,47.00\n"
                   "R,Primer_0_R,TTTCCGTGTAAAC,0.38,40.00,1.00,39.99\n")   # f32(39.995) = 39.99499...


@pytest.mark.gpu
def test_coverage_report_text(host, oracle):
    """main.rs:518-594: the per-segment search runs on the device (msspe_segment_coverage_dev), the
    totals and the text on the host; ragged record lengths and gap runs included."""
    import msspe_amd
    import ref_pipeline
    g = msspe_amd.synth.aligned_genomes(6, 1400)
    recs = [(f"s{i}", bytes(r).decode()) for i, r in enumerate(g)]
    recs[2] = (recs[2][0], recs[2][1][:1130])            # a shorter record: fewer partitions
    recs[4] = (recs[4][0], recs[4][1][:700] + "-" * 40 + recs[4][1][740:])
    segs = oracle.Segments([s for _, s in recs[:2]], 500, 250, 50, 13)
    fwd = [w for w, _ in segs.candidates(0, 3, 1)]
    rev = [w for w, _ in segs.candidates(1, 2, 1)]
    for f, r in ((fwd, rev), (fwd, []), ([], rev)):
        want = ref_pipeline.coverage_report(f, r, recs, 500, 250, 50, 13)
        rc, out = call(host.odm_coverage_report, "\n".join(f"{n}\t{s}" for n, s in recs).encode(),
                       "\n".join(f).encode(), "\n".join(r).encode(), 500, 250, 50, 13)
        assert rc > 0 and out == want


def test_coverage_report_needs_a_gpu(host):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    rc, out = call(host.odm_coverage_report, b"a\tACGT", b"ACGT", b"", 4, 2, 4, 2)
    assert rc == -2 and out


def test_cli_needs_a_gpu_for_the_pipeline(host, tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    fa = tmp_path / "in.fa"
    fa.write_text(">a\n" + "ACGT" * 200 + "\n")
    rc, out = call(host.odm_run_cli, *argv("-i", str(fa), "-o", str(tmp_path / "o.csv"), "--do-align", "false"))
    assert rc == 1 and "no CPU fallback" in out


@pytest.mark.gpu
@pytest.mark.parametrize("extra,kw", [
    ([], {}),
    (["--check-self-dimers", "false", "--max-iterations", "40"], dict(check_self_dimers=False, max_iterations=40)),
    (["--keep-all", "true", "--max-mismatch-segments", "3"], dict(keep_all=True, max_mismatch_segments=3)),
    (["--delta-g-threshold", "-4000", "--annealing-temp", "37", "--disable-tm-stddev", "true"],
     dict(dg=-4000.0, temp=37.0, disable_tm_stddev=True)),
    # 20-mers: stage C runs on the split-table kernel (thal_pairs_split.hip)
    (["--kmer-size", "20", "--max-tm", "80", "--max-iterations", "60"], dict(kmer_size=20, max_tm=80.0, max_iterations=60)),
])
def test_end_to_end_cli_matches_the_restated_pipeline(host, tmp_path, extra, kw):
    """Whole run (stage A -> B -> C -> vertex cover -> CSV + report) on a synthetic alignment:
    the CSV and the report must equal the oracle-based restatement of main.rs byte for byte."""
    import msspe_amd
    import ref_pipeline
    g = msspe_amd.synth.aligned_genomes(24, 3200)
    fasta = "".join(f">genome{i} synthetic\n{bytes(r).decode()}\n" for i, r in enumerate(g))
    fa, csv = tmp_path / "in.fa", tmp_path / "out.csv"
    fa.write_text(fasta)
    rc, report = call(host.odm_run_cli, *argv("-i", str(fa), "-o", str(csv), "--do-align", "false", *extra))
    assert rc == 0, report
    want_csv, want_report, info = ref_pipeline.run(fasta, **kw)
    assert csv.read_text() == want_csv
    assert report == want_report
    assert want_csv.count("\n") > 3


def _fake_mafft(tmp_path, body):
    bindir = tmp_path / "bin"
    bindir.mkdir()
    exe = bindir / "mafft"
    exe.write_text("#!/bin/sh\n" + body)
    exe.chmod(0o755)
    return bindir


def test_default_invocation_runs_mafft_with_the_reference_argv(host, tmp_path, monkeypatch):
    """od-msspe/src/main.rs:127-146 + config.rs:131-138: --do-align defaults to "true" and the input goes
    through `mafft --auto --quiet --thread -1 --op 1.53 --ep 0.123 --jtt 200 <file>`; its stdout is the
    FASTA.  A fake mafft records its argv and emits nothing, so the run ends in the reference's next
    panic ("No sequences found"), before any GPU call."""
    log = tmp_path / "argv.txt"
    bindir = _fake_mafft(tmp_path, f'printf "%s\\n" "$@" > {log}\n')
    monkeypatch.setenv("PATH", f"{bindir}:/usr/bin:/bin")
    fa = tmp_path / "in.fa"
    fa.write_text(">a\nACGT\n")
    rc, out = call(host.odm_run_cli, *argv("-i", str(fa), "-o", str(tmp_path / "o.csv")))
    assert rc == 101 and out == "No sequences found in the input file"
    assert log.read_text().split("\n")[:-1] == ["--auto", "--quiet", "--thread", "-1", "--op", "1.53", "--ep",
                                                "0.123", "--jtt", "200", str(fa)]


def test_mafft_output_is_the_alignment_that_gets_screened(host, tmp_path, monkeypatch):
    """The aligned FASTA on mafft's stdout replaces the input file (here: a fake mafft that prints a
    different file); without a GPU the run then stops at the engine, with one it completes."""
    import torch
    aligned = tmp_path / "aligned.fa"
    aligned.write_text(">x\n" + "ACGT" * 200 + "\n>y\n" + "ACGT" * 200 + "\n")
    bindir = _fake_mafft(tmp_path, f"cat {aligned}\n")
    monkeypatch.setenv("PATH", f"{bindir}:/usr/bin:/bin")
    fa = tmp_path / "in.fa"
    fa.write_text(">unaligned\nAC\n")
    rc, out = call(host.odm_run_cli, *argv("-i", str(fa), "-o", str(tmp_path / "o.csv")))
    if torch.cuda.is_available():
        assert rc == 0, out
    else:
        assert rc == 1 and "no CPU fallback" in out      # it got past the FASTA stage with mafft's records


def test_missing_mafft_panics_like_the_reference(host, tmp_path, monkeypatch):
    monkeypatch.setenv("PATH", str(tmp_path))           # no mafft anywhere
    fa = tmp_path / "in.fa"
    fa.write_text(">a\nACGT\n")
    rc, out = call(host.odm_run_cli, *argv("-i", str(fa), "-o", str(tmp_path / "o.csv"), "--do-align", "true"))
    assert rc == 101
    assert out == 'failed to execute MAFFT: Os { code: 2, kind: NotFound, message: "No such file or directory" }'


def test_format_ntthal_input_golden_and_skip_rules(host, golden_dir):
    """od-msspe/src/delta_g.rs:162-193 (test_format_ntthal_input) through the host's mirror of
    format_ntthal_input, plus the two skip rules of delta_g.rs:64-73 the reference does not test."""
    import json
    import pyoracle
    g = json.loads((golden_dir / "ntthal_format.json").read_text())
    primers = "\n".join(g["primers"]).encode()
    rc, out = call(host.odm_format_ntthal_input, primers, int(g["check_cross_dimers"]), int(g["check_self_dimers"]))
    assert rc >= 0 and out == g["expected"]
    assert out.split("\n") == [f"{a},{b}" for a in g["primers"] for b in g["primers"]]    # row-major, a outer
    # check_cross_dimers = false: nothing at all is sent, self pairs included (delta_g.rs:71-73)
    assert call(host.odm_format_ntthal_input, primers, 0, 1)[1] == ""
    # check_self_dimers = false: a == b and a == revcomp(b) are skipped (delta_g.rs:66-69)
    a = g["primers"][0]
    trio = [a, pyoracle.reverse_complement(a), g["primers"][1]]
    rc, out = call(host.odm_format_ntthal_input, "\n".join(trio).encode(), 1, 0)
    want = [f"{x},{y}" for x in trio for y in trio if not (x == y or pyoracle.reverse_complement(y) == x)]
    assert out.split("\n") == want and len(want) == 4
