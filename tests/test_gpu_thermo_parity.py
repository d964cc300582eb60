"""Parity of the HIP path (through the C-ABI) against the CPU oracle and the golden vectors.

Bar: bit-exact doubles for dS/dH/dG/t (both sides evaluate the same IEEE-754 operations in the
same order, no FMA), identical conflict decisions, identical counts/bitmaps.
"""
import json

import numpy as np
import pytest

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


def bitmap_to_bool(bm, n):
    bits = np.unpackbits(bm.view(np.uint8), axis=1, bitorder="little")
    return bits[:, :n].astype(bool)


def check_pool(eng, m, oracle, oracle_tables, pool, chem_kw=None, threshold=-9000.0):
    chem_kw = chem_kw or {}
    chem = m.Chem.ntthal(**chem_kw)
    oargs = oracle.ntthal_args(**{k: v for k, v in chem_kw.items()})
    out = eng.cross_dimer(pool, chem, threshold, want_dg=True, want_tm=True)
    cnt, dg, cf, tt = oracle.pool_pairs(oracle_tables, pool, oargs, threshold, want_t=True)
    n = len(pool)
    np.testing.assert_array_equal(out["dg"], dg)
    np.testing.assert_array_equal(out["tm"], tt)
    got = bitmap_to_bool(out["bitmap"], n)
    np.testing.assert_array_equal(got, cf.astype(bool))
    np.testing.assert_array_equal(out["row_conflicts"], cf.sum(1).astype(np.uint32))
    return out, cnt


def test_dimer_goldens_through_the_c_abi(eng, m, golden_dir):
    """od-msspe/src/delta_g.rs:196-230 via msspe_cross_dimer."""
    g = json.loads((golden_dir / "ntthal_dimer.json").read_text())
    for v in g["vectors"]:
        pool = [v["oligo1"], v["oligo2"]]
        out = eng.cross_dimer(pool, m.Chem.ntthal(temp_c=v["temp_c"]), 100000.0, want_tm=True)
        assert "%g" % out["dg"][0, 1] == v["dG"], v["id"]
        assert "%g" % out["tm"][0, 1] == v["t"], v["id"]
        assert out["row_conflicts"].tolist() == [2, 2]      # threshold 100000 keeps every pair


def test_random_pool_bit_exact(eng, m, oracle, oracle_tables):
    pool = m.synth.pool_strings(m.synth.random_pool(160, 13))
    check_pool(eng, m, oracle, oracle_tables, pool)


def test_chemistry_and_threshold_variants(eng, m, oracle, oracle_tables):
    pool = m.synth.pool_strings(m.synth.random_pool(48, 13, seed=5))
    check_pool(eng, m, oracle, oracle_tables, pool, dict(temp_c=37.0), -2000.0)
    check_pool(eng, m, oracle, oracle_tables, pool, dict(mv=100.0, dv=0.0, dntp=0.0, dna_conc=50.0), -1000.0)
    check_pool(eng, m, oracle, oracle_tables, pool, dict(dv=1.5, dntp=0.6), -3000.0)


def test_loop_size_limit_on_short_oligos(eng, m, oracle, oracle_tables):
    """thal.c maxLoop below 2k - 4: the register-table kernels have no cut-off, the split-table kernel
    (and behind it the one-wave-per-pair kernel) takes the block."""
    pool = m.synth.pool_strings(m.synth.random_pool(120, 13, seed=9))
    for max_loop in (0, 3, 8):
        chem = m.Chem.ntthal()
        chem.max_loop = max_loop
        out = eng.cross_dimer(pool, chem, -9000.0, want_dg=True, want_tm=True)
        _, dg, cf, tt = oracle.pool_pairs(oracle_tables, pool, oracle.ntthal_args(max_loop=max_loop), -9000.0, want_t=True)
        np.testing.assert_array_equal(out["dg"], dg)
        np.testing.assert_array_equal(out["tm"], tt)
        np.testing.assert_array_equal(out["row_conflicts"], cf.sum(1).astype(np.uint32))
    assert eng.pair_stage_stats()["deferred"] > 0      # the integer stage ran


def test_edge_pools(eng, m, oracle, oracle_tables):
    """No-structure pairs, maximal DP tables (poly-A x poly-T: 169 cells), homopolymers, repeats."""
    pool = ["A" * 13, "T" * 13, "C" * 13, "G" * 13, "ACACACACACACA", "TGTGTGTGTGTGT",
            "AAAAAAATTTTTT", "GGGGGGGCCCCCC", "AGCCCGTGTAAAC", "GTTTACACGGGCT", "ATATATATATATA"]
    out, _ = check_pool(eng, m, oracle, oracle_tables, pool)
    assert np.isinf(out["dg"][0, 0]) and out["tm"][0, 0] == 0.0          # poly-A vs poly-A
    assert np.isfinite(out["dg"][0, 1])


@pytest.mark.parametrize("k", [18, 24, 31])
def test_edge_pools_long_oligos(eng, m, oracle, oracle_tables, k):
    """Low-complexity long oligos: tables of k^2 cells (poly-A x poly-T: beyond every LDS table, dense
    kernel), dinucleotide repeats (hundreds of exact ties), self-complementary pairs, near-duplicates."""
    rep = lambda unit: (unit * k)[:k]
    half = rep("ACGGT")[:k // 2]
    pool = ["A" * k, "T" * k, "G" * k, "C" * k, rep("AC"), rep("GT"), rep("TG"), rep("AT"), rep("GC"),
            rep("AAT"), rep("ATT"), rep("ACG"), rep("CGT"), rep("AACCGGTT"), rep("AAAACCCCGGGGTTTT"),
            half + oracle.reverse_complement(half) + ("A" if k % 2 else ""), rep("GCGCAT"), rep("ATGCGC")]
    pool += m.synth.pool_strings(m.synth.random_pool(6, k, seed=k))
    pool.append(pool[-1][:-1] + ("A" if pool[-1][-1] != "A" else "C"))
    out, _ = check_pool(eng, m, oracle, oracle_tables, pool)
    assert np.isinf(out["dg"][0, 0]) and np.isfinite(out["dg"][0, 1])


@pytest.mark.parametrize("k", [6, 8, 12, 16, 17, 20, 21, 24, 27, 28, 29, 32])
def test_other_oligo_lengths(eng, m, oracle, oracle_tables, k):
    """k <= 14: register-table kernels; 15 .. 32: the table split over 2, 4 or 8 lanes
    (thal_pairs_split.hip) with the one-wave-per-pair f64 kernel behind it (thal_pairs_wave.hip)."""
    pool = m.synth.pool_strings(m.synth.random_pool(24 if k <= 16 else 40, k, seed=100 + k))
    if k % 2 == 0:   # both-self-complementary pairs use the symmetric concentration term
        half = pool[0][:k // 2]
        pal = half + oracle.reverse_complement(half)
        pool += [pal, ("GC" * k)[:k]]
    check_pool(eng, m, oracle, oracle_tables, pool)


@pytest.mark.parametrize("k,oligos,chem_kw,thr", [
    (14, ["GCGGCGGCCGCCGC", "GCCGGCCGGGCGGG", "GGCCGGCCGGGCGG"], dict(temp_c=37.0), -6000.0),
    (16, ["TCTAGACTAGCCAGCA", "TGAAGAAAGCTAAGTC"], dict(mv=200.0, dv=0.5, dntp=0.2), -1500.0)])
def test_a_resolved_pick_whose_walks_meet_a_tie_stays_open(eng, m, oracle, oracle_tables, k, oligos, chem_kw, thr):
    """Two cells tie in the terminal pick and the list stage settles them by comparing the doubles of both walks
    (thal_pairs_int.hip RESOLVE, thal_pairs_split.hip): if either walk passes a cell with an equal-valued alternative
    its last bits need not be thal()'s, so the comparison decides nothing and the pair belongs to the f64 kernels.
    These pairs came out 15e-12 cal/mol off / with the other cell's structure (pair campaign, seed 301, round 3);
    all orders of them, among random oligos of the same length."""
    pool = oligos + m.synth.pool_strings(m.synth.random_pool(150, k, seed=7 * k))
    for pair_kernel in ("auto", "int"):
        eng.set_option("pair_kernel", pair_kernel)
        try:
            check_pool(eng, m, oracle, oracle_tables, pool, chem_kw, thr)
        finally:
            eng.set_option("pair_kernel", "auto")


@pytest.mark.parametrize("lane_from", [None, 0])
def test_oligo_stats_bit_exact(eng, m, oracle, oracle_tables, golden_dir, lane_from):
    """primer3_core view (od-msspe/src/primer.rs:143-166): Tm, GC%, SELF_ANY/END, HAIRPIN.  lane_from = 0: the
    self-dimers through the one-lane-per-oligo kernels large pools take (one fill, ANY and END1 picks)."""
    if lane_from is not None:
        eng.set_option("self_lane_from", lane_from)
    try:
        _oligo_stats_bit_exact(eng, m, oracle, oracle_tables, golden_dir)
    finally:
        eng.set_option("self_lane_from", 81920)


def _oligo_stats_bit_exact(eng, m, oracle, oracle_tables, golden_dir):
    pool = m.synth.pool_strings(m.synth.random_pool(300, 13, seed=9))
    pool += ["AGCCCGTGTAAAC", "ACGTGAAAACGTA", "GCGCTTTTGCGCA", "GGGGCCCTTTGGG", "ATATATATATATA",
             "GGGGGGGCCCCCC", "AAAAAAAAAAAAA", "CCCGGGAAACCCG",
             # END1 corner cases: no partner for the 3' base anywhere (last row of the DP empty), with
             # and without other base pairs; a 3' base whose only partner is the 5' base
             "AAAAAAAAAAAAC", "GGGGGGGGGGGCA", "TTTTTTTTTTTTA", "TAAAAAAAAAAAA", "CACACACACACAG"]
    got = eng.oligo_stats(pool)
    ref = oracle.check_primers(oracle_tables, pool)
    for a, b in (("tm", "tm"), ("gc", "gc"), ("self_any", "self_any_th"),
                 ("self_end", "self_end_th"), ("hairpin", "hairpin_th")):
        np.testing.assert_array_equal(got[a], ref[b], err_msg=a)
    g = json.loads((golden_dir / "primer3_check_primers.json").read_text())["check_primers"][0]
    i = pool.index(g["primer"])
    assert m.round_fixed_f32(got["tm"][i], 3) == np.float32(g["tm"])
    assert m.round_fixed_f32(got["gc"][i], 3) == np.float32(g["gc"])
    assert got["self_any"][i] == 0.0 and got["self_end"][i] == 0.0 and got["hairpin"][i] == 0.0
    assert (got["hairpin"] > 0).sum() >= 3 and (got["self_any"] > 0).sum() >= 3


@pytest.mark.parametrize("k,n,lane_from", [(13, 90000, None), (16, 33000, 0), (9, 35000, 0)])
def test_self_dimers_of_a_large_pool_one_lane_per_oligo(eng, m, oracle, oracle_tables, k, n, lane_from):
    """From 81,920 oligos per call (option self_lane_from) SELF_ANY / SELF_END run one lane per oligo (f64 register
    tables, 56 then 72 slots, the wave kernel and the dense kernel behind them): same doubles as the oracle, whichever
    of the two is asked for, and as the one-wave-per-oligo path."""
    import torch
    if lane_from is not None:
        eng.set_option("self_lane_from", lane_from)
    rng = np.random.default_rng(k * 1000 + 7)
    pool = m.synth.random_pool(n, k, seed=77 + k)
    comp = {65: 84, 67: 71, 71: 67, 84: 65}
    for j in range(0, n, 17):            # self-complementary oligos (dense kernel) and near-palindromes (large tables)
        half = pool[j, :k // 2].copy()
        pool[j, k - k // 2:] = np.array([comp[c] for c in half[::-1]], dtype=np.uint8)
    pool[5] = np.frombuffer(b"ATATATATATATATATAT"[:k], dtype=np.uint8)       # tables beyond 72 cells
    pool[6] = np.frombuffer(b"GCGCGCGCGCGCGCGCGC"[:k], dtype=np.uint8)
    pool[7] = np.frombuffer(b"AAAAAAAAAAAAAAAAAC"[-k:], dtype=np.uint8)      # END1: empty last row
    words = m.synth.pool_strings(pool)
    ref = oracle.check_primers(oracle_tables, words)
    got = eng.oligo_stats(words)
    np.testing.assert_array_equal(got["self_any"], ref["self_any_th"])
    np.testing.assert_array_equal(got["self_end"], ref["self_end_th"])
    assert (got["self_any"] > 0).sum() > n // 100 and (got["self_end"] > 0).sum() > n // 200
    # each statistic asked for alone (one finish per fill), on the device-pointer entry point
    d_pool = torch.from_numpy(m.pack_oligos(pool).view(np.int64)).cuda()
    d_out = torch.full((2, n), -1.0, dtype=torch.float64, device="cuda")
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    try:
        eng.oligo_stats_dev(d_pool.data_ptr(), n, k, m.Chem.primer3(), d_self_any=d_out[0].data_ptr())
        eng.oligo_stats_dev(d_pool.data_ptr(), n, k, m.Chem.primer3(), d_self_end=d_out[1].data_ptr())
        torch.cuda.synchronize()
        alone = d_out.cpu().numpy()
        np.testing.assert_array_equal(alone[0], ref["self_any_th"])
        np.testing.assert_array_equal(alone[1], ref["self_end_th"])
        eng.set_option("self_lane_from", 1 << 30)            # the one-wave-per-oligo path on the same pool
        d_out.fill_(-1.0)
        eng.oligo_stats_dev(d_pool.data_ptr(), n, k, m.Chem.primer3(), d_self_any=d_out[0].data_ptr(),
                            d_self_end=d_out[1].data_ptr())
        torch.cuda.synchronize()
        wave = d_out.cpu().numpy()
        np.testing.assert_array_equal(wave[0], ref["self_any_th"])
        np.testing.assert_array_equal(wave[1], ref["self_end_th"])
    finally:
        eng.set_option("self_lane_from", 81920)
        eng.reset_stream()


def test_oligo_stats_longer_oligos(eng, m, oracle, oracle_tables):
    pool = m.synth.pool_strings(m.synth.random_pool(40, 20, seed=21)) + ["GGGGCCCTTTTGGGCCCCAA"]
    got = eng.oligo_stats(pool)
    ref = oracle.check_primers(oracle_tables, pool)
    for a, b in (("tm", "tm"), ("self_any", "self_any_th"), ("self_end", "self_end_th"),
                 ("hairpin", "hairpin_th")):
        np.testing.assert_array_equal(got[a], ref[b], err_msg=a)


def test_argument_errors(eng, m):
    with pytest.raises(m.MsspeError) as e:
        eng.cross_dimer(["ACGTNCGTACGTA"])
    assert e.value.code == 1
    with pytest.raises(m.MsspeError):
        eng.cross_dimer(["A"])          # k = 1 unsupported
    assert eng.cross_dimer([])["row_conflicts"].size == 0


def test_counts_and_bitmap_only_path_medium_pool(eng, m, oracle, oracle_tables):
    """512^2 ordered pairs through the device-pointer entry point without the dense dG output
    (the configuration bench.py times), on the caller's stream, several tiles per block."""
    import torch
    n = 512
    pool_ascii = m.synth.random_pool(n, 13, seed=77)
    d_pool = torch.from_numpy(m.pack_oligos(pool_ascii).view(np.int64)).cuda()
    d_rc = torch.zeros(n, dtype=torch.int32, device="cuda")
    d_bm = torch.zeros((n, (n + 63) // 64), dtype=torch.int64, device="cuda")
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    try:
        for _ in range(2):          # twice: results must not depend on leftovers of the last call
            d_rc.zero_()
            eng.cross_dimer_dev(d_pool.data_ptr(), n, 13, m.Chem.ntthal(), -9000.0, (0, n), (0, n),
                                d_rc.data_ptr(), d_bm.data_ptr())
        torch.cuda.synchronize()
    finally:
        eng.reset_stream()
    cnt, _, cf, _ = oracle.pool_pairs(oracle_tables, pool_ascii, want_dg=False)
    got = bitmap_to_bool(d_bm.cpu().numpy().view(np.uint64), n)
    np.testing.assert_array_equal(got, cf.astype(bool))
    np.testing.assert_array_equal(d_rc.cpu().numpy().astype(np.int64), cf.sum(1))
    assert int(d_rc.sum()) == cnt
    assert eng.last_overflow_pairs() > 0      # the overflow stages were exercised


@pytest.mark.parametrize("k,kw,thr", [
    (13, {}, -2500.0), (13, {}, -6000.0), (13, dict(temp_c=60.0, dv=0.0), -500.0),
    (13, dict(temp_c=37.0, mv=100.0, dv=1.5, dntp=0.2, dna_conc=50.0), -3000.0),
    (13, dict(temp_c=10.0, mv=1500.0, dv=0.0), -6000.0),      # positive salt term
    (11, {}, -2000.0), (15, dict(temp_c=45.0), -2500.0), (16, {}, -3500.0)])
def test_decisions_without_planes_equal_the_exact_planes(eng, m, oracle, oracle_tables, k, kw, thr):
    """A call without dG / Tm planes lets a pair with a tied terminal pick stand as "no conflict" when no
    structure of the tie could reach the cut (int_core.hpp kPickMargin: enthalpy range of the tied cells x
    (1 - T / 310.15), salt term x the pairs a structure can have); with planes every tie is settled exactly.
    Both calls must flag the same pairs, at thresholds inside the bulk of the dG distribution (a fifth to a
    half of all pairs conflict, so thousands of ties sit near the cut), for temperatures on both sides of
    37 C and both signs of the salt term; the planes themselves are compared with the oracle on a sample."""
    n = 3072
    pool_ascii = m.synth.random_pool(n, k, seed=1000 + k)
    pool = m.synth.pool_strings(pool_ascii)
    chem = m.Chem.ntthal(**kw)
    exact = eng.cross_dimer(pool, chem, thr, want_dg=True, want_tm=True)
    eng.pair_stage_stats()
    fast = eng.cross_dimer(pool, chem, thr, want_dg=False, want_tm=False)
    np.testing.assert_array_equal(fast["bitmap"], exact["bitmap"])
    np.testing.assert_array_equal(fast["row_conflicts"], exact["row_conflicts"])
    frac = float(exact["row_conflicts"].sum()) / (n * n)
    assert 0.02 < frac < 0.9, frac                      # the cut is inside the distribution
    # the bitmap of the exact call is its own dG plane cut the reference's way
    dec = bitmap_to_bool(exact["bitmap"], n)
    np.testing.assert_array_equal(dec, exact["dg"] <= m.g_cut(thr))
    # ... and the plane is the oracle's, on a block of rows
    _, dg, cf, _ = oracle.pool_pairs(oracle_tables, pool_ascii, oracle.ntthal_args(**kw), thr, rows=(0, 48))
    np.testing.assert_array_equal(exact["dg"][:48], dg)
    np.testing.assert_array_equal(dec[:48], cf.astype(bool))


@pytest.mark.parametrize("k", [13, 15])
def test_decisions_at_cuts_that_are_pairs_own_values(eng, m, k):
    """A decisions-only call takes a pair's dG from the sums the integer DP carries (two roundings) and replays the
    structure in the reference's own order of f64 additions only when a lane of the wave comes within 1e-3 cal/mol
    of the cut (thal_pairs_row.hip kCutMargin).  Thresholds that ARE the dG of pairs of the pool, as the f32 the
    reference compares in, put pairs exactly there: every decision must still be the exact plane's."""
    n = 1536
    pool = m.synth.pool_strings(m.synth.random_pool(n, k, seed=77 + k))
    chem = m.Chem.ntthal()
    exact = eng.cross_dimer(pool, chem, -9000.0, want_dg=True)
    dg = exact["dg"]
    finite = np.sort(dg[np.isfinite(dg)])
    near = 0
    for q in (0.002, 0.01, 0.05, 0.2, 0.5):
        thr = float(np.float32(finite[int(q * finite.size)]))
        cut = m.g_cut(thr)
        near += int((np.abs(dg - cut) < 1e-3).sum())
        fast = eng.cross_dimer(pool, chem, thr, want_dg=False, want_tm=False)
        np.testing.assert_array_equal(bitmap_to_bool(fast["bitmap"], n), dg <= cut)
        np.testing.assert_array_equal(fast["row_conflicts"], (dg <= cut).sum(1).astype(np.uint32))
    assert near > 0      # pairs sat inside the margin


@pytest.mark.parametrize("k,pair_kernel", [(9, "auto"), (13, "auto"), (13, "int"), (13, "f64"), (15, "auto")])
def test_pairs_without_a_complementary_cell(eng, m, oracle, oracle_tables, k, pair_kernel):
    """Two-letter pools (T/C against T/C, A/G against A/G): whole waves in which no pair has a single
    complementary cell.  thal() finds no structure there (dG = inf, t = 0) and the planes must say so -- the
    integer kernels used to leave such a wave's plane entries unwritten (found by a randomised campaign with
    skewed compositions; decisions and counts were never affected)."""
    rng = np.random.default_rng(31 + k)
    tc = np.frombuffer(b"TC", dtype=np.uint8)[rng.integers(0, 2, (160, k))]
    ag = np.frombuffer(b"AG", dtype=np.uint8)[rng.integers(0, 2, (130, k))]
    mixed = m.synth.random_pool(60, k, seed=5)
    skew = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.choice(4, size=(150, k), p=[0.05, 0.45, 0.05, 0.45])]
    pool = m.synth.pool_strings(np.concatenate([tc, ag, mixed, skew]))
    eng.set_option("pair_kernel", pair_kernel)
    try:
        out, cnt = check_pool(eng, m, oracle, oracle_tables, pool)
    finally:
        eng.set_option("pair_kernel", "auto")
    assert np.isinf(out["dg"][:160, :160]).all() and np.isinf(out["dg"][160:290, 160:290]).all()
    assert cnt > 0


@pytest.mark.parametrize("split_list", [1, 0])
def test_very_large_tables_of_short_oligos(eng, m, oracle, oracle_tables, split_list):
    """A/T-only and G/C-only 13-mers against each other: 60 ... 169 complementary cells per pair, far beyond the
    integer stages' tables.  With split_list = 1 they go through the split-table kernel's list mode (two lanes
    per pair, up to 128 cells; all-A against all-T is left to the one-wave-per-pair kernel), with 0 straight
    to the f64 kernels: same planes, bit for bit, as the oracle's."""
    rng = np.random.default_rng(44)
    at = np.frombuffer(b"AT", dtype=np.uint8)[rng.integers(0, 2, (110, 13))]
    gc = np.frombuffer(b"GC", dtype=np.uint8)[rng.integers(0, 2, (90, 13))]
    homo = np.frombuffer(b"AAAAAAAAAAAAATTTTTTTTTTTTTGGGGGGGGGGGGGCCCCCCCCCCCCC", dtype=np.uint8).reshape(4, 13)
    pool = m.synth.pool_strings(np.concatenate([at, gc, homo, m.synth.random_pool(52, 13, seed=9)]))
    eng.set_option("split_list", split_list)
    try:
        out, cnt = check_pool(eng, m, oracle, oracle_tables, pool, threshold=-6000.0)
        fast = eng.cross_dimer(pool, m.Chem.ntthal(), -6000.0, want_dg=False, want_tm=False)
    finally:
        eng.set_option("split_list", 1)
    np.testing.assert_array_equal(fast["bitmap"], out["bitmap"])
    assert cnt > 1000


def test_hand_over_lists_shrink_when_the_card_is_full(m):
    """A 32,768-primer screen (2^30 pairs) asks for two hand-over lists of 8 GB; with only 10 GB left on the
    card the engine takes smaller lists (more flushes) and the screen comes out the same."""
    import torch
    n = 32768
    pool_ascii = m.synth.random_pool(n, 13, seed=909)
    d_pool = torch.from_numpy(m.pack_oligos(pool_ascii).view(np.int64)).cuda()
    counts = []
    for squeeze in (False, True):
        eng = m.Engine(0)
        d_rc = torch.zeros(n, dtype=torch.int32, device="cuda")
        hog = None
        try:
            if squeeze:
                torch.cuda.empty_cache()
                free, _ = torch.cuda.mem_get_info()
                hog = torch.empty(max(free - (10 << 30), 0), dtype=torch.uint8, device="cuda")
            eng.cross_dimer_dev(d_pool.data_ptr(), n, 13, m.Chem.ntthal(), -9000.0, (0, n), (0, n), d_rc.data_ptr())
            eng.synchronize()
            counts.append(d_rc.cpu().numpy().copy())
            assert eng.last_overflow_pairs() > 0
        finally:
            del hog
            eng.close()
            torch.cuda.empty_cache()
    np.testing.assert_array_equal(counts[0], counts[1])
    assert int(counts[0].sum()) > 0


@pytest.mark.parametrize("k,no_split", [(13, False), (21, False), (21, True)])
def test_row_and_column_sub_blocks(eng, m, oracle, oracle_tables, monkeypatch, k, no_split):
    """A rectangular block of the pair matrix (what one rank computes in the multi-GPU tiling); for
    21-mers through the split-table kernel and, with that one switched off, through the matrix mode
    of the one-wave-per-pair kernel."""
    import torch
    eng.set_option("split_min_k", 99 if no_split else 16)
    n = 200
    pool_ascii = m.synth.random_pool(n, k, seed=78)
    d_pool = torch.from_numpy(m.pack_oligos(pool_ascii).view(np.int64)).cuda()
    r0, r1, c0, c1 = 37, 150, 64, 200
    d_rc = torch.zeros(n, dtype=torch.int32, device="cuda")
    d_dg = torch.full((r1 - r0, c1 - c0), 7.0, dtype=torch.float64, device="cuda")
    d_bm = torch.zeros((r1 - r0, (c1 - c0 + 63) // 64), dtype=torch.int64, device="cuda")
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    try:
        eng.cross_dimer_dev(d_pool.data_ptr(), n, k, m.Chem.ntthal(), -9000.0, (r0, r1), (c0, c1),
                            d_rc.data_ptr(), d_bm.data_ptr(), d_dg.data_ptr())
        torch.cuda.synchronize()
    finally:
        eng.reset_stream()
        eng.set_option("split_min_k", 16)
    _, dg, cf, _ = oracle.pool_pairs(oracle_tables, pool_ascii)
    np.testing.assert_array_equal(d_dg.cpu().numpy(), dg[r0:r1, c0:c1])
    want = np.zeros(n, dtype=np.int64)
    want[r0:r1] = cf[r0:r1, c0:c1].sum(1)
    np.testing.assert_array_equal(d_rc.cpu().numpy().astype(np.int64), want)
    got = bitmap_to_bool(d_bm.cpu().numpy().view(np.uint64), c1 - c0)
    np.testing.assert_array_equal(got, cf[r0:r1, c0:c1].astype(bool))


def test_large_pool_properties(eng, m, oracle, oracle_tables, monkeypatch):
    """16,384^2 = 2.7e8 ordered pairs (16 launches; the hand-over lists are forced down to 2^27
    entries so that the stages behind the first one run twice mid-screen): size-independent
    properties of the counts/bitmap outputs plus an oracle check of sampled rows."""
    import torch
    eng.set_option("list_cap_log2", 27)
    n = 16384
    pool_ascii = m.synth.random_pool(n, 13)
    d_pool = torch.from_numpy(m.pack_oligos(pool_ascii).view(np.int64)).cuda()
    d_rc = torch.zeros(n, dtype=torch.int32, device="cuda")
    d_bm = torch.zeros((n, n // 64), dtype=torch.int64, device="cuda")
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    try:
        eng.cross_dimer_dev(d_pool.data_ptr(), n, 13, m.Chem.ntthal(), -9000.0, (0, n), (0, n),
                            d_rc.data_ptr(), d_bm.data_ptr())
        torch.cuda.synchronize()
    finally:
        eng.reset_stream()
        eng.set_option("list_cap_log2", 0)
    rc = d_rc.cpu().numpy().astype(np.int64)
    bits = np.unpackbits(d_bm.cpu().numpy().view(np.uint8), axis=1, bitorder="little")
    # (1) the per-row counts are the popcounts of the bitmap rows
    np.testing.assert_array_equal(bits.sum(1), rc)
    # (2) duplicates in the pool behave identically (rows of equal primers are equal)
    _, first, inv = np.unique(pool_ascii, axis=0, return_index=True, return_inverse=True)
    dup = np.flatnonzero(first[inv.ravel()] != np.arange(n))
    for i in dup[:8]:
        np.testing.assert_array_equal(bits[i], bits[first[inv.ravel()[i]]])
    # (3) dG(a,b) and dG(b,a) are numerically the same structure almost always: the conflict
    #     relation is nearly symmetric (the reference still evaluates both orders)
    asym = np.count_nonzero(bits != bits.T)
    assert asym <= 1e-4 * bits.sum() + 4
    # (4) sampled rows against the oracle
    rows = np.random.default_rng(9).choice(n, 12, replace=False)
    for r in rows:
        _, _, cf, _ = oracle.pool_pairs(oracle_tables, pool_ascii, rows=(int(r), int(r) + 1), want_dg=False)
        np.testing.assert_array_equal(bits[r], cf[0])
    assert eng.last_overflow_pairs() > 0


def test_integer_stage_equals_f64_stage(eng, m, oracle, oracle_tables, monkeypatch):
    """The exact-integer first stage (default) and the f64 register-table kernel
    (option pair_kernel = f64) must produce the same doubles and the same decisions; the integer
    stage may hand only a small share of the pairs on, for the documented reasons."""
    pool = m.synth.pool_strings(m.synth.random_pool(700, 13, seed=77))
    chem = m.Chem.ntthal()
    eng.pair_stage_stats()
    a = eng.cross_dimer(pool, chem, -9000.0, want_dg=True, want_tm=True)
    stats = eng.pair_stage_stats()
    eng.set_option("pair_kernel", "f64")
    try:
        b = eng.cross_dimer(pool, chem, -9000.0, want_dg=True, want_tm=True)
        assert eng.pair_stage_stats()["deferred"] == 0          # the integer stage did not run
        # the general integer kernel (what 14..16-mers and the list mode run) as the first stage
        eng.set_option("pair_kernel", "int")
        c = eng.cross_dimer(pool, chem, -9000.0, want_dg=True, want_tm=True)
        stats_general = eng.pair_stage_stats()
    finally:
        eng.set_option("pair_kernel", "auto")
    for key in ("dg", "tm", "bitmap", "row_conflicts"):
        np.testing.assert_array_equal(a[key], b[key])
        np.testing.assert_array_equal(a[key], c[key])
    # Which pairs a first stage answers itself differs between the two kernels (the row kernel pads a wave's rows
    # to its widest lane, the general one counts a lane's own cells), so their own counters differ.  What
    # the integers cannot settle is a property of the pair, though: every pair with such a reason reaches the
    # list stage whichever kernel ran first, so the list stage's counters must agree exactly ...
    assert stats_general["deferred"] > 0 and stats_general["replay_mismatch"] == 0
    assert stats["list"] == stats_general["list"] and stats["needed_f64"] == stats_general["needed_f64"]
    # ... and a pair both first stages ran and handed on was handed on for the same reasons (pick_tie, loop_tie,
    # path_tie, rejected_min, ...: bit for bit)
    small = pool[:120]
    eng.pair_stage_stats()
    eng.cross_dimer(small, chem, -9000.0, want_dg=True)
    by_row = {(r, c): bits for r, c, bits in eng.pair_stage_samples()}
    eng.pair_stage_stats()
    eng.set_option("pair_kernel", "int")
    try:
        eng.cross_dimer(small, chem, -9000.0, want_dg=True)
        by_int = {(r, c): bits for r, c, bits in eng.pair_stage_samples()}
        eng.pair_stage_stats()
    finally:
        eng.set_option("pair_kernel", "auto")
    both = set(by_row) & set(by_int)
    assert len(by_row) < 1024 and len(by_int) < 1024 and len(both) > 20
    assert all(by_row[pr] == by_int[pr] for pr in both)
    # identical runs give identical counters (the columns are grouped by a stable sort: the lane a pair runs
    # in, and with it every hand-over decision, is the same every time)
    eng.cross_dimer(pool, chem, -9000.0, want_dg=True, want_tm=True)
    assert eng.pair_stage_stats() == stats
    eng.cross_dimer(pool, chem, -9000.0)
    once = eng.pair_stage_stats()
    eng.cross_dimer(pool, chem, -9000.0)
    assert eng.pair_stage_stats() == once
    n2 = len(pool) ** 2
    assert 0 < stats["deferred"] < 0.06 * n2                  # retried in list mode
    assert 0 < stats["needed_f64"] < 0.01 * n2                # what only the f64 kernels can answer
    assert stats["needed_f64"] < stats["deferred"]            # the list mode settles two-cell picks
    assert stats["replay_mismatch"] == 0 and stats["tm_near_tie"] < 0.001 * n2
    assert stats["list"]["replay_mismatch"] == 0
    samples = eng.pair_stage_samples()
    assert all(0 <= r < len(pool) and 0 <= c < len(pool) and bits for r, c, bits in samples)
    # a few of the pairs that were handed on, against the oracle directly
    oargs = oracle.ntthal_args()
    for r, c, _bits in samples[:25]:
        res = oracle.thal(oracle_tables, pool[r], pool[c], oracle.ANY, oargs)
        if res.no_structure:
            assert np.isinf(a["dg"][r, c])
        else:
            assert a["dg"][r, c] == res.dG and a["tm"][r, c] == res.t


def test_row_kernel_probe_and_its_fallback(eng, m):
    """The row-specialised first stage reads LDS beyond its allocation and takes the 0 gfx950 returns there for
    "not available" (thal_pairs_row.hip).  The per-engine probe must find that behaviour on this device (or the
    headline numbers are the general kernel's), and the fallback it guards -- option row_oob = 0, the same
    branch a failed probe takes -- must give the same doubles and decisions and be the general integer kernel
    to the last counter."""
    assert eng.info("lds_reads_zero") == 1 and eng.info("row_kernel") == 1
    assert eng.info("n_cu") >= 64
    pool = m.synth.pool_strings(m.synth.random_pool(600, 13, seed=4242))
    chem = m.Chem.ntthal()
    eng.pair_stage_stats()
    a = eng.cross_dimer(pool, chem, -9000.0, want_dg=True, want_tm=True)
    stats_row = eng.pair_stage_stats()
    fast = eng.cross_dimer(pool, chem, -9000.0)
    eng.pair_stage_stats()
    try:
        eng.set_option("row_oob", 0)
        assert eng.info("row_kernel") == 0
        b = eng.cross_dimer(pool, chem, -9000.0, want_dg=True, want_tm=True)
        stats_fallback = eng.pair_stage_stats()
        fast_b = eng.cross_dimer(pool, chem, -9000.0)
        eng.pair_stage_stats()
        eng.set_option("row_oob", 1)
        eng.set_option("pair_kernel", "int")
        eng.cross_dimer(pool, chem, -9000.0, want_dg=True, want_tm=True)
        stats_int = eng.pair_stage_stats()
    finally:
        eng.set_option("row_oob", 1)
        eng.set_option("pair_kernel", "auto")
    for key in ("dg", "tm", "bitmap", "row_conflicts"):
        np.testing.assert_array_equal(a[key], b[key])
    for key in ("bitmap", "row_conflicts"):
        np.testing.assert_array_equal(fast[key], fast_b[key])
        np.testing.assert_array_equal(fast[key], a[key])
    assert stats_fallback == stats_int            # the fallback IS the general integer kernel
    assert stats_fallback != stats_row            # ... and the default is not
    with pytest.raises(m.MsspeError):
        eng.set_option("row_oob", 2)
    with pytest.raises(m.MsspeError):
        eng.info("no_such_key")


def test_small_fixed_lists_bound_the_launches(eng, m):
    """option list_cap_log2 below one default launch (2^27 pairs): launches shrink to what a list holds, so a list
    cannot be overrun whatever share of the pairs is handed on (here: every pair with planes, a pool of two
    letters whose tables are all oversized)."""
    pools = [m.synth.pool_strings(m.synth.random_pool(1500, 13, seed=5)),
             ["".join(np.random.default_rng(s).choice(list("AT"), 13)) for s in range(1300)]]
    chem = m.Chem.ntthal()
    for pool in pools:
        want = eng.cross_dimer(pool, chem, -9000.0)
        try:
            eng.set_option("list_cap_log2", 20)          # 2^20 entries < 1500^2 pairs
            got = eng.cross_dimer(pool, chem, -9000.0)
            assert eng.last_overflow_pairs() > 0
        finally:
            eng.set_option("list_cap_log2", 0)
        for key in ("bitmap", "row_conflicts"):
            np.testing.assert_array_equal(got[key], want[key])


def _read_bundle(path):
    """{section: [lines]} of a parameter bundle ('@ name count' headers)."""
    sections, name = {}, None
    for line in path.read_text().splitlines():
        if line.startswith("#") or not line.strip():
            continue
        if line.startswith("@"):
            name = line.split()[1]
            sections[name] = []
        else:
            sections[name].append(line)
    return sections


def _perturbed(sections, step_s, step_h):
    """Shift every available entry of the stack / mismatch / terminal-stack / dangle entropies by
    a multiple of step_s and the stack enthalpies by a multiple of step_h (tables stay plausible:
    the point is that the engine computes with whatever the files hold)."""
    out = {}
    for name, lines in sections.items():
        step = {"stack.ds": step_s, "stackmm.ds": step_s, "tstack2.ds": step_s, "tstack_tm_inf.ds": step_s,
                "dangle.ds": step_s, "stack.dh": step_h}.get(name)
        if step is None:
            out[name] = list(lines)
            continue
        q, new = 0, []
        for line in lines:
            toks = []
            for t in line.split():
                if t != "inf":
                    q += 1
                    t = repr(round(float(t) + step * (q % 3 - 1), 6))
                toks.append(t)
            new.append(" ".join(toks))
        out[name] = new
    return out


@pytest.mark.parametrize("k", [13, 20])
@pytest.mark.parametrize("mode", ["directory", "grid", "offgrid"])
def test_parameter_files_from_a_path(m, oracle, tmp_path, mode, k):
    """ntthal is run with `-path <primer3_config>/` (od-msspe/src/delta_g.rs:93-110): tables come from
    files.  directory = the stock tables split into Primer3's 16 files; grid = shifted values that
    still sit on the 0.01 cal/K grid (integer first stage, other numbers); offgrid = values off that
    grid, for which the integer stage must stand down and the f64 kernels answer alone (13 bases: the
    register-table f64 chain; 20 bases: one wave per pair, thal_pairs_wave.hip)."""
    sections = _read_bundle(oracle.default_bundle())
    if mode == "grid":
        sections = _perturbed(sections, 0.1, 100.0)
    elif mode == "offgrid":
        sections = _perturbed(sections, 0.003, 0.0)
    if mode == "directory":
        path = tmp_path / "primer3_config"
        path.mkdir()
        for name, lines in sections.items():
            (path / name).write_text("\n".join(lines) + "\n")
    else:
        path = tmp_path / "custom.bundle"
        path.write_text("# test bundle\n" + "".join(
            f"@ {name} {sum(len(l.split()) for l in lines)}\n" + "\n".join(lines) + "\n"
            for name, lines in sections.items()))
    tables = oracle.Tables(path)
    e = m.Engine(0, params_path=str(path))
    try:
        pool = m.synth.pool_strings(m.synth.random_pool(200, k, seed=77))
        out, _ = check_pool(e, m, oracle, tables, pool)
        stats = e.pair_stage_stats()
        if mode == "offgrid":
            assert stats["deferred"] == 0 and stats["pick_tie"] == 0      # integer stage not used
        else:
            assert stats["deferred"] > 0                                  # integer stage ran (and met ties)
        if mode != "directory":     # the perturbed tables really give other numbers
            ref = oracle.pool_pairs(oracle.Tables(), pool)[1]
            assert (out["dg"] != ref).mean() > 0.5
    finally:
        e.close()


@pytest.mark.parametrize("k,max_loop", [(18, 30), (20, 30), (22, 6), (26, 30), (32, 30)])
def test_long_oligo_stage_equals_generic_kernel(m, oracle, oracle_tables, monkeypatch, k, max_loop):
    """The split-table integer stage (15 .. 32 bases) against the dense f64 kernel on a pool the
    oracle would need minutes for: same decisions, same counts; a sample of rows bit-exact against
    the oracle; a loop-size limit below 30 is honoured (thal.c maxLoop)."""
    n = 768
    pool = m.synth.pool_strings(m.synth.random_pool(n, k, seed=4000 + k))
    # a few designed cases: perfect duplexes, a self-complementary pair, homopolymers
    pool[1] = oracle.reverse_complement(pool[0])
    half = pool[2][:k // 2]
    pool[3] = half + oracle.reverse_complement(half)
    pool[4], pool[5] = "A" * k, "T" * k
    chem = m.Chem.ntthal()
    chem.max_loop = max_loop
    e = m.Engine(0)
    try:
        fast = e.cross_dimer(pool, chem, -9000.0)
        stats = e.pair_stage_stats()
        handed_on = e.last_overflow_pairs()
        e.set_option("force_generic", 1)
        slow = e.cross_dimer(pool, chem, -9000.0)
        e.set_option("force_generic", 0)
        np.testing.assert_array_equal(fast["bitmap"], slow["bitmap"])
        np.testing.assert_array_equal(fast["row_conflicts"], slow["row_conflicts"])
        assert fast["row_conflicts"].sum() > 0
        assert 0 < handed_on < 0.05 * n * n          # the integer stage answered nearly everything
        assert stats["replay_mismatch"] == 0 and stats["tm_near_tie"] == 0
        rows = (0, 2, 4, 5, 300, n - 1)
        sub = e.cross_dimer([pool[r] for r in rows] + pool, chem, -9000.0, want_dg=True, want_tm=True)
        oargs = oracle.ntthal_args(max_loop=max_loop)
        for q, r in enumerate(rows):
            _, dg, cf, tt = oracle.pool_pairs(oracle_tables, [pool[r]] + pool, oargs, -9000.0, want_t=True,
                                             rows=(0, 1))
            np.testing.assert_array_equal(sub["dg"][q, len(rows):], dg[0, 1:])
            np.testing.assert_array_equal(sub["tm"][q, len(rows):], tt[0, 1:])
    finally:
        e.close()


def test_edge_list_output(m, oracle, oracle_tables):
    """msspe_cross_dimer_edges: (a, b, dG) per conflicting ordered pair, sorted like the reference's nested
    loops emit them, dG = what Edge::get_dg() returns ("%g" -> f32 -> "{:.2}" -> f32, delta_g.rs:10-15,
    33-46); a capacity that is too small is reported with the count needed."""
    pool = m.synth.pool_strings(m.synth.random_pool(400, 13, seed=5))
    thr = -7000.0
    eng = m.Engine(0)
    try:
        edges, count = eng.cross_dimer_edges(pool, m.Chem.ntthal(), thr, capacity=1 << 16)
        dense = eng.cross_dimer(pool, m.Chem.ntthal(), thr, want_dg=True)
        with pytest.raises(m.MsspeError) as e:
            eng.cross_dimer_edges(pool, m.Chem.ntthal(), thr, capacity=7)
        assert e.value.code == 5 and e.value.count == count           # MSSPE_ERR_CAPACITY + edges needed
        assert np.count_nonzero(e.value.edges["a"] | e.value.edges["b"]) >= 6    # the first 7 are filled in
    finally:
        eng.close()
    _, dg, cf, _ = oracle.pool_pairs(oracle_tables, pool, oracle.ntthal_args(), thr)
    want = np.argwhere(cf.astype(bool))                                # row-major = sorted by (a, b)
    assert count == len(want) == int(dense["row_conflicts"].sum()) and count > 100
    np.testing.assert_array_equal(np.stack([edges["a"], edges["b"]], 1), want)
    for q in range(0, count, 7):
        a, b = want[q]
        first = oracle.round_g_f32(float(dg[a, b]))
        assert edges["dg"][q] == np.float32(oracle.round_fixed_f32(float(first), 2))
        assert edges["dg"][q] < np.float32(thr)


@pytest.mark.parametrize("k,n", [(13, 6000), (16, 1500), (20, 1200), (24, 800), (28, 500), (32, 400)])
def test_hairpin_wave_kernel_bit_exact(m, oracle, oracle_tables, k, n):
    """HAIRPIN_TH (primer.rs:104-106 -> Primer3 thal type 4) from the wave-per-oligo kernel with its planes
    in LDS: 10,400 oligos over six lengths, random ones plus designed stem-loops (a stem of 4..7 pairs
    around loops of 3..6 bases at several offsets), bit-exact against the oracle and against the
    one-lane kernel over a global workspace (option force_generic)."""
    rng = np.random.default_rng(1000 + k)
    pool = m.synth.pool_strings(m.synth.random_pool(n, k, seed=3000 + k))
    for q in range(n // 4):                                   # every fourth oligo: a designed hairpin
        stem = int(rng.integers(4, 8))
        loop = int(rng.integers(3, 7))
        if 2 * stem + loop > k:
            stem = (k - loop) // 2
        left = "".join("ACGT"[x] for x in rng.integers(0, 4, stem))
        mid = "".join("ACGT"[x] for x in rng.integers(0, 4, loop))
        core = left + mid + oracle.reverse_complement(left)
        pad = k - len(core)
        off = int(rng.integers(0, pad + 1))
        flank = "".join("ACGT"[x] for x in rng.integers(0, 4, pad))
        pool[4 * q] = flank[:off] + core + flank[off:]
    eng = m.Engine(0)
    try:
        got = eng.oligo_stats(pool)["hairpin"]
        eng.set_option("force_generic", 1)
        slow = eng.oligo_stats(pool)["hairpin"]
    finally:
        eng.close()
    ref = oracle.check_primers(oracle_tables, pool)["hairpin_th"]
    np.testing.assert_array_equal(got, ref)
    np.testing.assert_array_equal(got, slow)
    assert (got > 0).sum() >= n // 8                           # the designed ones do fold
