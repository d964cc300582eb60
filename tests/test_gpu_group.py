"""Several devices behind the C ABI (include/msspe_hip.h msspe_group_*): the pair matrix of
od-msspe/src/delta_g.rs:61-81 dealt out over a group of contexts, the pool assembled by one all-gather, the counts
merged by one all-reduce.  One card is all a test box has, so the members share device 0 (the collectives then run
as device copies: the same code path above the two-function collective layer); RCCL itself is exercised as a group of
one rank in a child process (it refuses two ranks per device).  Results must equal the single-context calls."""
import ctypes as C
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


@pytest.fixture(scope="module")
def m():
    import msspe_amd
    return msspe_amd


@pytest.fixture(scope="module")
def eng(m):
    e = m.Engine(0)
    yield e
    e.close()


@pytest.mark.parametrize("members,n", [(3, 1700), (2, 100), (4, 2051), (1, 300)])
def test_group_on_one_card_equals_one_context(m, eng, members, n):
    pool = m.synth.pool_strings(m.synth.random_pool(n, 13, seed=100 + n))
    chem = m.Chem.ntthal()
    want = eng.cross_dimer(pool, chem, -8000.0, want_dg=False)
    want_edges, want_count = eng.cross_dimer_edges(pool, chem, -8000.0)
    want_stats = eng.oligo_stats(pool)
    g = m.Group([0] * members)
    try:
        assert g.size == members and g.transport == ("device-copy" if members > 1 else "single")
        got = g.cross_dimer(pool, chem, -8000.0)
        np.testing.assert_array_equal(got["row_conflicts"], want["row_conflicts"])
        np.testing.assert_array_equal(got["bitmap"], want["bitmap"])
        counts_only = g.cross_dimer(pool, chem, -8000.0, want_bitmap=False)
        np.testing.assert_array_equal(counts_only["row_conflicts"], want["row_conflicts"])
        edges, count = g.cross_dimer_edges(pool, chem, -8000.0)
        assert count == want_count and count > 0
        np.testing.assert_array_equal(edges, want_edges)
        # the capacity contract of the single-context call: too small -> MSSPE_ERR_CAPACITY and the count needed
        with pytest.raises(m.MsspeError) as err:
            g.cross_dimer_edges(pool, chem, -8000.0, capacity=max(1, count // 3))
        assert err.value.count == count
        stats = g.oligo_stats(pool)
        for key in want_stats:
            np.testing.assert_array_equal(stats[key], want_stats[key])
        # a second screen of another size on the same group (buffers are reused or grown)
        pool2 = m.synth.pool_strings(m.synth.random_pool(n + 333, 13, seed=7))
        np.testing.assert_array_equal(g.cross_dimer(pool2, chem, -9000.0)["row_conflicts"],
                                      eng.cross_dimer(pool2, chem, -9000.0, want_dg=False)["row_conflicts"])
    finally:
        g.close()


def test_group_options_and_refusals(m):
    with pytest.raises(m.MsspeError):
        m.Group([0, 0], transport="rccl")            # RCCL needs one rank per device
    with pytest.raises(m.MsspeError):
        m.Group([0], transport="smoke-signals")
    with pytest.raises(m.MsspeError):
        m.Group([])
    with pytest.raises(m.MsspeError):
        m.Group([99])                                 # no such device
    g = m.Group([0, 0])
    try:
        g.set_option("pair_kernel", "int")           # reaches every member
        with pytest.raises(m.MsspeError):
            g.set_option("pair_kernel", "abacus")
        with pytest.raises(m.MsspeError):
            g.cross_dimer(["ACGTNACGTACGT"] * 4)
    finally:
        g.close()


_RCCL_CHILD = r"""
import sys
sys.path.insert(0, sys.argv[1])
import numpy as np
import msspe_amd as m
pool = m.synth.pool_strings(m.synth.random_pool(900, 13, seed=5))
chem = m.Chem.ntthal()
e = m.Engine(0)
want = e.cross_dimer(pool, chem, -8500.0, want_dg=False)
g = m.Group([0], transport="rccl")
assert g.transport == "rccl", g.transport
got = g.cross_dimer(pool, chem, -8500.0)
assert np.array_equal(got["row_conflicts"], want["row_conflicts"]) and np.array_equal(got["bitmap"], want["bitmap"])
g.close(); e.close()
print("rccl-one-rank-ok")
"""


def test_rccl_transport_with_one_rank():
    """librccl.so is loaded, a communicator made and the all-gather / all-reduce issued through it (one rank: the
    only RCCL configuration one card allows).  In a child process: a collective that hung would not take the suite
    with it."""
    out = subprocess.run([sys.executable, "-c", _RCCL_CHILD, str(ROOT / "open-msspe-design_amd")], capture_output=True,
                         text=True, timeout=300)
    assert out.returncode == 0 and "rccl-one-rank-ok" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]


def test_cli_with_devices_writes_the_single_device_csv(m, tmp_path):
    g = m.synth.aligned_genomes(60, 5200, seed=11)
    fasta = "".join(f">g{i} synthetic\n{bytes(r).decode()}\n" for i, r in enumerate(g))
    fa = tmp_path / "in.fa"
    fa.write_text(fasta)
    m.load_library()
    host = C.CDLL(str(ROOT / "open-msspe-design_amd" / "libod_msspe_host.so"))
    outs = {}
    for tag, extra in (("one", []), ("group", ["--devices", "0,0,0"]), ("group1", ["--devices", "0"])):
        csv = tmp_path / f"{tag}.csv"
        args = ["od-msspe-hip", "-i", str(fa), "-o", str(csv), "--do-align", "false", "--delta-g-threshold", "-6500",
                "--max-iterations", "200"] + extra
        arr = (C.c_char_p * len(args))(*[a.encode() for a in args])
        buf = C.create_string_buffer(1 << 20)
        assert host.odm_run_cli(len(args), arr, buf, 1 << 20) == 0, buf.value.decode()
        outs[tag] = (csv.read_text(), buf.value.decode())
    assert outs["one"] == outs["group"] == outs["group1"]
    assert outs["one"][0].count("\n") > 10
    bad = ["od-msspe-hip", "-i", str(fa), "-o", str(tmp_path / "x.csv"), "--devices", "0,x", "--do-align=false"]
    arr = (C.c_char_p * len(bad))(*[a.encode() for a in bad])
    buf = C.create_string_buffer(1 << 16)
    assert host.odm_run_cli(len(bad), arr, buf, 1 << 16) == 2


@pytest.mark.parametrize("transport", ["rccl", "device-copy", None])
def test_group_on_two_distinct_devices_equals_one_context(m, eng, transport):
    """Distinct devices: grouped ncclAllGather / ncclAllReduce on N communicators from one thread, peer access and
    cross-device copies, the summing kernel on the root -- what one card cannot execute.  Skipped on a one-GPU box
    (every box the build has had so far: these paths are NOT YET VERIFIED ON HARDWARE)."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs in one node")
    n = 3000
    pool = m.synth.pool_strings(m.synth.random_pool(n, 13, seed=4321))
    chem = m.Chem.ntthal()
    want = eng.cross_dimer(pool, chem, -8500.0, want_dg=False)
    want_edges, want_count = eng.cross_dimer_edges(pool, chem, -8500.0)
    g = m.Group([0, 1], transport=transport)
    try:
        assert g.size == 2 and g.transport == (transport or "rccl"), (g.transport, g.transport_reason)
        got = g.cross_dimer(pool, chem, -8500.0)
        np.testing.assert_array_equal(got["row_conflicts"], want["row_conflicts"])
        np.testing.assert_array_equal(got["bitmap"], want["bitmap"])
        edges, count = g.cross_dimer_edges(pool, chem, -8500.0)
        assert count == want_count
        np.testing.assert_array_equal(edges, want_edges)
        # a pool with fewer rows than members x 256: the second member screens nothing and must still be in step
        small = pool[:200]
        np.testing.assert_array_equal(g.cross_dimer(small, chem, -8500.0)["row_conflicts"],
                                      eng.cross_dimer(small, chem, -8500.0, want_dg=False)["row_conflicts"])
        e2, c2 = g.cross_dimer_edges(small, chem, -8500.0)
        w2, wc2 = eng.cross_dimer_edges(small, chem, -8500.0)
        assert c2 == wc2
        np.testing.assert_array_equal(e2, w2)
        stats = g.oligo_stats(pool[:500])
        ref = eng.oligo_stats(pool[:500])
        for key in ref:
            np.testing.assert_array_equal(stats[key], ref[key])
    finally:
        g.close()


def test_auto_transport_reports_why_it_runs_the_copies(m):
    """Members that share a card cannot use RCCL: "auto" takes the copies without a reason to report (RCCL was never
    wanted), an explicit "rccl" is refused."""
    g = m.Group([0, 0])
    try:
        assert g.transport == "device-copy" and g.transport_reason == ""
    finally:
        g.close()
    with pytest.raises(m.MsspeError):
        m.Group([0, 0], transport="rccl")
