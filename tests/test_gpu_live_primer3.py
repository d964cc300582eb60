"""Cross-check against a LIVE Primer3 when the GPU box has one (SURVEY.md 7 "Hard parts", 8c, 8d).

The reference computes nothing itself: Tm / GC / SELF_ANY / SELF_END / HAIRPIN come from `primer3_core`
(od-msspe/src/primer.rs:143-166) and dG from `ntthal -i` (od-msspe/src/delta_g.rs:83-153).  Neither binary
is in this image, so END1, hairpin and positive SELF_ANY values are pinned by no reference vector
(DESIGN.md 2).  These tests run the real executables as child processes when `ntthal` / `primer3_core` are
on $PATH and compare them with the C-ABI outputs; otherwise they skip (and say so)."""
import shutil
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

NTTHAL = shutil.which("ntthal")
PRIMER3 = shutil.which("primer3_core")


@pytest.fixture(scope="module")
def m():
    import msspe_amd
    return msspe_amd


@pytest.mark.skipif(NTTHAL is None, reason="no ntthal on $PATH: thal ANY stays pinned by delta_g.rs:196-230 only")
def test_cross_dimer_dg_against_live_ntthal(m):
    n = 320                                    # 102,400 ordered pairs, one ntthal process (-i reads them all)
    pool = m.synth.pool_strings(m.synth.random_pool(n, 13, seed=4242))
    eng = m.Engine(0)
    try:
        out = eng.cross_dimer(pool, m.Chem.ntthal(), -9000.0, want_dg=True)
    finally:
        eng.close()
    stdin = "\n".join(f"{a},{b}" for a in pool for b in pool)
    res = subprocess.run([NTTHAL, "-a", "ANY", "-mv", "50.00", "-dv", "3.00", "-n", "0.00", "-d", "250.00",
                          "-t", "25.00", "-i"], input=stdin, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[:500]
    lines = res.stdout.splitlines()
    q = 0
    bad = 0
    for i in range(n):
        for j in range(n):
            if np.isinf(out["dg"][i, j]):      # no structure: ntthal prints nothing for the pair
                continue
            tok = lines[5 * q].split()
            q += 1
            # north star: within 0.01 kcal/mol; the text has 6 significant digits
            if abs(float(tok[13]) - out["dg"][i, j]) > max(0.5, 5e-6 * abs(out["dg"][i, j])):
                bad += 1
    assert q * 5 == len(lines) and bad == 0


@pytest.mark.skipif(PRIMER3 is None, reason="no primer3_core on $PATH: END1 / hairpin / SELF_ANY > 0 stay unpinned")
def test_oligo_stats_against_live_primer3_core(m):
    n = 1000
    pool = m.synth.pool_strings(m.synth.random_pool(n, 13, seed=777))
    pool[:4] = ["ACGTGAAAACGTA", "GCGCTTTTGCGCA", "GGGCCCTTTGGGC", "AGCCCGTGTAAAC"]     # designed stem-loops + D1
    eng = m.Engine(0)
    try:
        got = eng.oligo_stats(pool)
    finally:
        eng.close()
    rec = "".join(f"SEQUENCE_ID={p}\nSEQUENCE_PRIMER={p}\nPRIMER_TASK=check_primers\nPRIMER_MIN_SIZE=13\n"
                  f"PRIMER_MIN_TM=29.00\nPRIMER_MAX_TM=59.00\nPRIMER_OPT_TM=59.00\nPRIMER_PICK_ANYWAY=1\n=\n"
                  for p in pool)               # od-msspe/src/primer.rs:125-140
    res = subprocess.run([PRIMER3], input=rec, capture_output=True, text=True, timeout=600)
    vals = {}
    cur = None
    for line in res.stdout.splitlines():
        if line.startswith("SEQUENCE_ID="):
            cur = line.split("=", 1)[1]
            vals[cur] = {}
        elif cur and line.startswith("PRIMER_LEFT_0_"):
            k, v = line.split("=", 1)
            vals[cur][k] = v
    keys = {"tm": ("PRIMER_LEFT_0_TM", 3), "gc": ("PRIMER_LEFT_0_GC_PERCENT", 3),
            "self_any": ("PRIMER_LEFT_0_SELF_ANY_TH", 2), "self_end": ("PRIMER_LEFT_0_SELF_END_TH", 2),
            "hairpin": ("PRIMER_LEFT_0_HAIRPIN_TH", 2)}
    for i, p in enumerate(pool):
        for name, (tag, dec) in keys.items():
            want = np.float32(vals[p][tag])
            have = np.float32(m.round_fixed_f32(float(got[name][i]), dec))
            assert have == want, (p, name, have, want)


def test_the_probe_reports_what_it_found():
    """Always runs: leaves a line in the log saying whether the live cross-check happened."""
    print(f"live Primer3 probe: ntthal={NTTHAL!r} primer3_core={PRIMER3!r}")
