"""The long-oligo integer recurrence of csrc/thal_pairs_split.hip (17 .. 28 bases), restated in
plain Python (tests/split_dp_model.py) over the tables the kernel keeps in LDS (exported host-side by
msspe_host_split_tables), against the oracle's fillMatrix planes.  Runs without a GPU."""
import numpy as np
import pytest

import int_dp_model
import split_dp_model as model


@pytest.fixture(scope="module")
def tables():
    import msspe_amd
    return model.load_tables(msspe_amd)


@pytest.fixture(scope="module")
def consts():
    import msspe_amd
    tb = int_dp_model.load_tables(msspe_amd)
    return tb.init_S, tb.RC


def test_split_tables_are_usable_and_consistent(tables):
    tb = tables
    assert tb.usable and 24 <= tb.max_k <= 32
    big = model.K_BIG
    L = tb.L.reshape(32, 32)
    assert L[0, 0] == big                                   # the stacked pair is not a loop candidate
    assert np.all(L[0, 1:31] == 0) and np.all(L[1:31, 0] == 0) and L[1, 1] == 0
    for l1 in range(32):
        for l2 in range(32):
            if l1 + l2 > 30:
                assert L[l1, l2] == big                     # thal.c MAX_LOOP
    # interior loops: the two parts add up to the folded entry of the compact plane
    for (l1, l2) in ((1, 2), (2, 1), (5, 5), (3, 11), (14, 16), (1, 29)):
        sz = l1 + l2
        whole = tb.g[model.K_NB + (sz - 2) * 64: model.K_NB + (sz - 2) * 64 + 64].astype(np.int64)
        parts = int(L[l1, l2]) + tb.X[model.K_XP:model.K_XP + 64].astype(np.int64)
        ok = whole != big
        assert ok.sum() >= 48 and np.array_equal(parts[ok], whole[ok] + 600000 * abs(l1 - l2))
        assert np.all(tb.X[model.K_XP:model.K_XP + 64][~ok] == big)
    assert np.array_equal(tb.X[model.K_XMM:model.K_XMM + 64], tb.g[model.K_NB:model.K_NB + 64])   # 1 x 1
    for sz in (1, 2, 17, 30):
        for pe in range(16):
            want = tb.g[model.K_BU + (pe >> 2) * model.K_BUSTRIDE + sz * 4 + (pe & 3)]
            assert tb.X[model.K_XB1 + sz * 16 + pe] == want and tb.X[model.K_XB2 + sz * 16 + pe] == want


def test_max_loop_cuts_the_loop_rows():
    import msspe_amd
    tb = model.load_tables(msspe_amd, max_loop=8)
    L = tb.L.reshape(32, 32)
    assert L[4, 4] != model.K_BIG and L[4, 5] == model.K_BIG and L[0, 9] == model.K_BIG and L[0, 8] == 0


@pytest.mark.parametrize("k,n_pairs", [(18, 60), (24, 40), (28, 25)])
def test_recurrence_matches_oracle_planes(tables, consts, oracle, oracle_tables, k, n_pairs):
    import msspe_amd
    args = oracle.ntthal_args()
    rng = np.random.default_rng(k)
    pool = msspe_amd.synth.pool_strings(msspe_amd.synth.random_pool(64, k, seed=k))
    hard = 0
    for _ in range(n_pairs):
        a, b = pool[rng.integers(0, 64)], pool[rng.integers(0, 64)]
        cells, is_hard = model.run_pair(tables, consts[0], consts[1], a, b)
        S, H = oracle.dimer_planes(oracle_tables, a, b, args)
        S = np.asarray(S).reshape(k, k)
        H = np.asarray(H).reshape(k, k)
        assert len(cells) == int(np.isfinite(H).sum())
        if is_hard:
            hard += 1       # the kernel hands such a pair to the f64 kernel
            continue
        for (i, j), (G, Hc, _po) in cells.items():
            assert Hc == H[i, j], (a, b, i, j)
            assert abs((2000.0 * H[i, j] - 620300.0 * S[i, j]) - G) < 0.5, (a, b, i, j)
    assert hard < 0.2 * n_pairs
