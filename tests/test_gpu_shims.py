"""The protocol shims (bin/ntthal-hip, bin/primer3_core-hip): an unmodified od-msspe selects its
executables with --ntthal / --primer3 (config.rs:142-147); these speak the same pipes."""
import json
import subprocess
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
BIN = Path(__file__).resolve().parent.parent / "open-msspe-design_amd" / "bin"


def parse_ntthal_output(input_text: str, output: str, threshold: float):
    """delta_g.rs:27-59 restated: zip input lines with 5-line blocks, token 13 = dG."""
    edges = {}
    out_lines = iter(output.splitlines())
    for line in input_text.splitlines():
        first = next(out_lines, None)
        if first is not None:
            toks = first.split()
            if len(toks) > 13:
                dg = np.float32(toks[13])
                if dg < np.float32(threshold):
                    a, b = line.split(",")
                    edges[(a, b)] = "%.2f" % dg
        for _ in range(4):
            next(out_lines, None)
    return edges


def test_ntthal_shim_reproduces_the_reference_transcript(golden_dir):
    """The exact argv of delta_g.rs:93-110 and the transcript of delta_g.rs:206-230."""
    g = json.loads((golden_dir / "ntthal_dimer.json").read_text())
    for temp in (37.0, 25.0):
        vecs = [v for v in g["vectors"] if v["temp_c"] == temp]
        stdin = "\n".join(f'{v["oligo1"]},{v["oligo2"]}' for v in vecs)
        res = subprocess.run([str(BIN / "ntthal-hip"), "-a", "ANY", "-mv", "50.00", "-dv", "3.00", "-n", "0.00",
                              "-d", "250.00", "-t", "%.2f" % temp, "-path", "/nonexistent/primer3_config/", "-i"],
                             input=stdin, capture_output=True, text=True)
        # a missing -path directory is an error for ntthal as well
        assert res.returncode != 0
        res = subprocess.run([str(BIN / "ntthal-hip"), "-a", "ANY", "-mv", "50.00", "-dv", "3.00", "-n", "0.00",
                              "-d", "250.00", "-t", "%.2f" % temp, "-i"],
                             input=stdin, capture_output=True, text=True)
        assert res.returncode == 0, res.stderr
        lines = res.stdout.splitlines()
        assert len(lines) == 5 * len(vecs)
        for q, v in enumerate(vecs):
            head = lines[5 * q].split()
            assert head[:5] == ["Calculated", "thermodynamical", "parameters", "for", "dimer:"]
            assert (head[7], head[10], head[13], head[16]) == (v["dS"], v["dH"], v["dG"], v["t"])
            rows = [r.replace("\t", " " * v["tab_spaces"]).rstrip() for r in lines[5 * q + 1: 5 * q + 5]]
            assert rows == [d.rstrip() for d in v["drawing"]]
        edges = parse_ntthal_output(stdin, res.stdout, 100000.0)
        assert len(edges) == len(vecs)


def test_ntthal_shim_prints_nothing_for_a_pair_without_structure():
    res = subprocess.run([str(BIN / "ntthal-hip"), "-a", "ANY", "-t", "25", "-i"],
                         input="AAAAAAAAAAAAA,AAAAAAAAAAAAA\nAGGCCTATATCCA,GAAGCAGTATTTT",
                         capture_output=True, text=True)
    assert res.returncode == 0 and len(res.stdout.splitlines()) == 5


def test_primer3_shim_golden(golden_dir):
    """primer.rs:218-250: od-msspe's exact Boulder-IO record in, the five values it reads back out."""
    g = json.loads((golden_dir / "primer3_check_primers.json").read_text())
    res = subprocess.run([str(BIN / "primer3_core-hip")], input=g["format_input"]["expected"],
                         capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    kv = dict(l.split("=", 1) for l in res.stdout.splitlines() if "=" in l and l != "=")
    want = g["check_primers"][0]
    assert kv["SEQUENCE_ID"] == want["primer"]
    assert np.float32(kv["PRIMER_LEFT_0_TM"]) == np.float32(want["tm"])
    assert np.float32(kv["PRIMER_LEFT_0_GC_PERCENT"]) == np.float32(want["gc"])
    assert (kv["PRIMER_LEFT_0_SELF_ANY_TH"], kv["PRIMER_LEFT_0_SELF_END_TH"], kv["PRIMER_LEFT_0_HAIRPIN_TH"]) == \
        ("0.00", "0.00", "0.00")
    assert res.stdout.rstrip().endswith("=")


def test_ntthal_shim_answers_the_reference_input_text_in_order(golden_dir):
    """The text of delta_g.rs:162-193 (four ordered pairs, no trailing newline) piped into the shim with
    the reference's argv: one 5-line block per input line, in input order, so that parse_ntthal_output's
    zip (delta_g.rs:31-56) attributes every dG to the right pair."""
    import pyoracle
    g = json.loads((golden_dir / "ntthal_format.json").read_text())
    stdin = g["expected"]
    res = subprocess.run([str(BIN / "ntthal-hip"), "-a", "ANY", "-mv", "50.00", "-dv", "3.00", "-n", "0.00",
                          "-d", "250.00", "-t", "25.00", "-i"], input=stdin, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    lines = res.stdout.splitlines()
    pairs = [l.split(",") for l in stdin.split("\n")]
    assert len(lines) == 5 * len(pairs)
    tables = pyoracle.Tables()
    for q, (a, b) in enumerate(pairs):
        want = pyoracle.thal(tables, a, b, pyoracle.ANY, pyoracle.ntthal_args())
        assert lines[5 * q].split()[13] == "%g" % want.dG
