"""Oracle vs. the reference's golden vectors for the thermodynamic arithmetic (no GPU)."""
import json
import math

import numpy as np
import pytest

from helpers import draw_dimer


def g6(x: float) -> str:
    return "%g" % x


def test_tables_bundle_equals_primer3_dir(oracle):
    """The shipped bundle and a Primer3-format directory load to identical tables (only checkable
    where the reference tree exists, i.e. in the build container)."""
    from pathlib import Path
    ref = Path("/root/reference/od-msspe/primer3_config")
    if not ref.is_dir():
        pytest.skip("reference tree not present (GPU box)")
    import ctypes as C
    a, b = oracle.Tables(), oracle.Tables(ref)
    size = 4 * 2 * 625 * 8 + 4 * 125 * 8   # the dense tables at the head of orc_tables
    assert C.string_at(a.ptr, size) == C.string_at(b.ptr, size)


def test_dimer_goldens(oracle, oracle_tables, golden_dir):
    """od-msspe/src/delta_g.rs:196-230: dS dH dG t (as ntthal prints them, %g) and the drawing."""
    g = json.loads((golden_dir / "ntthal_dimer.json").read_text())
    c = g["conditions"]
    for v in g["vectors"]:
        args = oracle.ntthal_args(c["mv"], c["dv"], c["dntp"], c["dna_conc"], v["temp_c"],
                                  c["max_loop"])
        r = oracle.thal(oracle_tables, v["oligo1"], v["oligo2"], oracle.ANY, args)
        assert not r.no_structure
        assert (g6(r.dS), g6(r.dH), g6(r.dG), g6(r.t)) == (v["dS"], v["dH"], v["dG"], v["t"]), v["id"]
        n = len(v["oligo1"])
        pairs = [[i + 1, r.ps1[i]] for i in range(n) if r.ps1[i]]
        assert pairs == v["pairs"], v["id"]
        rows = [row.replace("\t", " " * v["tab_spaces"]).rstrip() for row in
                draw_dimer(v["oligo1"], v["oligo2"], list(r.ps1)[:n], list(r.ps2)[:n])]
        assert rows == [d.rstrip() for d in v["drawing"]], v["id"]


def test_dimer_dg_roundtrip_as_reference_parses_it(oracle, oracle_tables, golden_dir):
    """delta_g.rs:33-45,257-263: dG token -> f32 -> '{:.2}' -> f32 round-trips to the literals."""
    g = json.loads((golden_dir / "ntthal_dimer.json").read_text())
    c = g["conditions"]
    want = g["parse_expectations"]["edge_dg"]
    for v in g["vectors"]:
        key = f'{v["oligo1"]}:{v["oligo2"]}'
        if key not in want:
            continue
        args = oracle.ntthal_args(c["mv"], c["dv"], c["dntp"], c["dna_conc"], v["temp_c"])
        r = oracle.thal(oracle_tables, v["oligo1"], v["oligo2"], oracle.ANY, args)
        f = oracle.round_g_f32(r.dG)
        assert np.float32(float("%.2f" % f)) == np.float32(want[key])


def test_check_primers_golden(oracle, oracle_tables, golden_dir):
    """od-msspe/src/primer.rs:238-250 (primer3_core defaults 50/1.5/0.6/50)."""
    g = json.loads((golden_dir / "primer3_check_primers.json").read_text())
    for v in g["check_primers"]:
        i = oracle.check_primer(oracle_tables, v["primer"])
        assert i.tm_f32 == np.float32(v["tm"])
        assert i.gc_f32 == np.float32(v["gc"])
        assert i.self_any_f32 == np.float32(v["self_any_th"])
        assert i.self_end_f32 == np.float32(v["self_end_th"])
        assert i.hairpin_f32 == np.float32(v["hairpin_th"])
        assert "%.3f" % i.tm == "43.727" and "%.3f" % i.gc == "53.846"


def test_debug_checkpoints_from_survey(oracle, oracle_tables):
    """SURVEY.md C.3 scratch checkpoints (derived, not reference-published): RC, salt and a few
    DP cells under mv 50 / dv 3 / dNTP 0 / 250 nM."""
    S, H = oracle.dimer_planes(oracle_tables, "AGGCCTATATCCA", "GAAGCAGTATTTT")
    assert (round(S[2, 8], 4), H[2, 8]) == (-27.4, -9800.0)
    assert (round(S[3, 9], 4), H[3, 9]) == (-51.8, -19600.0)
    S, H = oracle.dimer_planes(oracle_tables, "CTGAAGCAGTATT", "GCATCTTTCCCTT")
    assert (round(S[5, 11], 4), H[5, 11]) == (-53.5, -18000.0)
    S, H = oracle.dimer_planes(oracle_tables, "AGTCCTGCGTGAT", "TGGCCTACATCAG")
    assert (round(S[9, 4], 4), H[9, 4]) == (-90.4773, -29100.0)


def test_no_structure_pair(oracle, oracle_tables):
    """poly-A vs poly-A has no complementary cell: ntthal prints nothing."""
    r = oracle.thal(oracle_tables, "A" * 13, "A" * 13)
    assert r.no_structure == 1 and r.t == 0.0
    conflict = oracle.lib().orc_pair_conflict(oracle_tables.ptr, b"A" * 13, b"A" * 13,
                                              oracle.ntthal_args(), -9000.0, None)
    assert conflict == 0


def test_g_rounding_rule(oracle):
    """-9000.004 prints as -9000 and is NOT below -9000."""
    assert oracle.round_g_f32(-9000.004) == -9000.0
    assert not (oracle.round_g_f32(-9000.004) < np.float32(-9000.0))
    assert oracle.round_g_f32(-9000.006) < np.float32(-9000.0)
    assert oracle.round_g_f32(-12345.67) == np.float32(-12345.7)


def test_dg_symmetry_is_only_approximate_so_both_orders_are_evaluated(oracle, oracle_tables):
    """The reference evaluates (a,b) and (b,a) separately (delta_g.rs:64-78); the oracle does too."""
    rng = np.random.default_rng(7)
    pool = ["".join("ACGT"[x] for x in rng.integers(0, 4, 13)) for _ in range(24)]
    cnt, dg, cf, _ = oracle.pool_pairs(oracle_tables, pool)
    fin = np.isfinite(dg) & np.isfinite(dg.T)
    assert np.allclose(dg[fin], dg.T[fin], atol=1e-6)
    assert cnt == int(cf.sum())


def test_hairpin_and_end1_are_restated_unpinned(oracle, oracle_tables):
    """No reference vector pins END1 or positive hairpin Tm; these checks only guard against
    regressions of the restatement on designed stem-loops (values recorded from this oracle)."""
    r = oracle.thal(oracle_tables, "GCGCTTTTGCGCA", "GCGCTTTTGCGCA", oracle.HAIRPIN, oracle.p3_args())
    assert not r.no_structure and r.n_pairs == 4 and 70.0 < r.t < 80.0
    r = oracle.thal(oracle_tables, "AGCCCGTGTAAAC", "AGCCCGTGTAAAC", oracle.HAIRPIN, oracle.p3_args())
    assert r.no_structure
    any_ = oracle.thal(oracle_tables, "GGGGCCCTTTTGGGCCCC", "GGGGCCCTTTTGGGCCCC", oracle.ANY, oracle.p3_args())
    end1 = oracle.thal(oracle_tables, "GGGGCCCTTTTGGGCCCC", "GGGGCCCTTTTGGGCCCC", oracle.END1, oracle.p3_args())
    assert math.isclose(any_.t, end1.t) and end1.end1 == 18
