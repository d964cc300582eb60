"""The C-ABI library loads and exports every symbol include/msspe_hip.h declares (no GPU)."""
import ctypes as C
import re
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


@pytest.fixture(scope="module")
def m():
    import msspe_amd
    return msspe_amd


def test_library_exports_every_declared_symbol(m):
    header = (ROOT / "include" / "msspe_hip.h").read_text()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = sorted(set(re.findall(r"\b(msspe_[a-z0-9_]+)\s*\(", header)))
    assert declared, "no prototypes found in include/msspe_hip.h"
    L = m.load_library()
    missing = [s for s in declared if not hasattr(L, s)]
    assert not missing, f"libmsspe_hip.so lacks: {missing}"
    from msspe_amd.capi import EXPORTS
    assert sorted(EXPORTS) == declared


def test_pack_unpack_and_rejects_non_acgt(m):
    w = m.pack_oligos(["ACGTACGTACGTA", "TTTTTTTTTTTTT", "AAAAAAAAAAAAA"])
    assert w.dtype == np.uint64 and w[2] == 0 and w[1] == (1 << 26) - 1
    assert m.unpack_oligo(int(w[0]), 13) == "ACGTACGTACGTA"
    with pytest.raises(m.MsspeError):
        m.pack_oligos(["ACGTNCGTACGTA"])
    with pytest.raises(m.MsspeError):
        m.pack_oligos(["A" * 33])


def test_text_rounding_matches_the_reference_rule(m, oracle):
    """SURVEY.md App. B; the product's rounding helpers agree with the oracle's."""
    rng = np.random.default_rng(3)
    for x in np.concatenate([rng.uniform(-20000, 2000, 2000), [-9000.004, -9000.006, -9999.995]]):
        assert m.round_g_f32(float(x)) == oracle.round_g_f32(float(x))
        assert m.round_fixed_f32(float(x) / 100, 3) == oracle.round_fixed_f32(float(x) / 100, 3)


@pytest.mark.parametrize("thr", [-9000.0, -2315.07, 100000.0, 0.0, -12345.6, 7.25, -500.0, -500.004, -87.125, 999.99])
def test_g_cut_is_the_exact_decision_boundary(m, oracle, thr):
    """The kernels test dG <= g_cut(threshold); that must be the reference's decision: the %g text as f32
    below the threshold (delta_g.rs:33-36) AND the "{:.2}" text of that f32 below it again (main.rs:758).
    Thresholds of magnitude < 1000 are where the second filter bites (%g keeps 3+ decimals there)."""
    cut = m.g_cut(thr)
    thr32 = float(np.float32(thr))
    assert oracle.edge_decision(cut, thr32)
    assert not oracle.edge_decision(float(np.nextafter(cut, np.inf)), thr32)
    rng = np.random.default_rng(11)
    for x in cut + rng.normal(0, max(1.0, abs(thr)) * 1e-4, 500):
        assert oracle.edge_decision(float(x), thr32) == (x <= cut)
    # the product's own helpers restate the same rule
    first = m.round_g_f32(cut)
    assert first < np.float32(thr) and m.round_fixed_f32(float(first), 2) < np.float32(thr)


def test_second_filter_drops_an_edge_the_first_one_keeps(m, oracle):
    """dG = -500.001 at threshold -500: %g prints -500.001 (< -500: kept by parse_ntthal_output), the
    stored "{:.2}" text is -500.00, which main.rs:758 does not count as below the threshold."""
    assert m.round_g_f32(-500.001) < np.float32(-500.0)
    assert not oracle.edge_decision(-500.001, -500.0)
    assert m.g_cut(-500.0) < -500.001
    assert oracle.edge_decision(-500.006, -500.0) and -500.006 <= m.g_cut(-500.0)


def test_create_fails_loudly_without_a_gpu(m):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(m.MsspeError) as e:
        m.Engine(0)
    assert e.value.code == 4 and "no CPU fallback" in str(e.value)


def test_bad_parameter_path_is_reported(m):
    with pytest.raises(m.MsspeError) as e:
        m.Engine(0, "/nonexistent/primer3_config/")
    assert e.value.code == 3
