"""The exact-integer recurrence of csrc/thal_pairs_int.hip, restated in plain Python
(tests/int_dp_model.py) over the very tables the kernel keeps in LDS (exported host-side by
msspe_host_pair_tables), against the oracle's fillMatrix planes: every cell's dG(37 C) * 2000 and
enthalpy must be the oracle's, as exact integers.  Runs without a GPU."""
import numpy as np
import pytest

import int_dp_model as model


@pytest.fixture(scope="module")
def tables():
    import msspe_amd
    return model.load_tables(msspe_amd)


def test_integer_tables_are_usable_and_consistent(tables):
    tb = tables
    assert tb.fast_ok and tb.int_ok
    big = 900000000
    finite = tb.H < (1 << 28)
    zt = np.zeros(model.K_COUNT, dtype=bool)
    zt[model.K_ZT:model.K_ZT + 64] = True
    # g = 20000 (H / 10) - 6203 round(100 S) wherever the entry exists; the asymmetry rows carry n
    s100 = np.rint(tb.S * 100.0)
    sel = finite & ~zt
    assert np.all(np.abs(tb.S[sel] * 100.0 - s100[sel]) < 1e-7)
    assert np.all(tb.H[sel] % 10 == 0)
    want = 20000 * (tb.H[sel].astype(np.int64) // 10) - 6203 * s100[sel].astype(np.int64)
    assert np.array_equal(tb.g[sel].astype(np.int64), want)
    assert np.all(tb.g[~finite] == big)
    d = np.arange(-32, 32)
    assert np.array_equal(tb.g[model.K_ZT:model.K_ZT + 64], 600000 * np.abs(d))
    # the stacked-pair row is void, and so is everything the geometry can never produce
    T = tb.T.reshape(model.K_ROWS, 64)
    assert np.all(T[0] == big)
    # an interior loop row = compact entry + 600000 |l1 - l2|
    for (l1, l2) in ((1, 2), (3, 1), (4, 4), (2, 7)):
        row = T[l1 * 16 + l2]
        src = tb.g[model.K_NB + (l1 + l2 - 2) * 64: model.K_NB + (l1 + l2 - 2) * 64 + 64].astype(np.int64)
        ok = src != big
        assert np.array_equal(row[ok], src[ok] + 600000 * abs(l1 - l2))
        assert np.all(row[~ok] == big)


def test_recurrence_matches_oracle_planes(tables, oracle, oracle_tables):
    import msspe_amd
    args = oracle.ntthal_args()
    rng = np.random.default_rng(20260630)
    pool = msspe_amd.synth.random_pool(300, 13)
    pool = [p if isinstance(p, str) else bytes(p).decode() for p in pool]
    deferred = 0
    n_pairs = 250
    for _ in range(n_pairs):
        a, b = pool[rng.integers(0, 300)], pool[rng.integers(0, 300)]
        cells, defer, _pick = model.run_pair(tables, a, b)
        S, H = oracle.dimer_planes(oracle_tables, a, b, args)
        S = np.asarray(S).reshape(13, 13)
        H = np.asarray(H).reshape(13, 13)
        assert len(cells) == int(np.isfinite(H).sum())
        if defer:
            deferred += 1   # exact tie somewhere: the f64 kernels answer these pairs
            continue
        for (i, j), (G, Hc, _pred, _po) in cells.items():
            assert Hc == H[i, j], (a, b, i, j)
            assert abs((2000.0 * H[i, j] - 620300.0 * S[i, j]) - G) < 0.5, (a, b, i, j)
    assert deferred < 0.15 * n_pairs


def test_known_strong_duplex(tables, oracle, oracle_tables):
    """A perfectly complementary pair: one helix, no loops, the model's terminal cell carries the
    full stack sum."""
    a = "ACGTTGCAAGGCT"
    b = oracle.reverse_complement(a)
    cells, defer, pick = model.run_pair(tables, a, b)
    S, H = oracle.dimer_planes(oracle_tables, a, b, oracle.ntthal_args())
    H = np.asarray(H).reshape(13, 13)
    assert pick == (12, 12) and cells[pick][1] == H[12, 12]
