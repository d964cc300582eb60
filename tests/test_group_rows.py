"""How the rows of the N^2 pair matrix (od-msspe/src/delta_g.rs:61-81) are dealt out over the members of a device
group (include/msspe_hip.h msspe_group_rows; host-only arithmetic, no GPU): groups of 256 rows, round robin."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def m():
    import msspe_amd
    return msspe_amd


@pytest.mark.parametrize("n,world", [(0, 2), (1, 2), (255, 2), (256, 2), (257, 2), (2000, 2), (65536, 8), (100003, 8),
                                     (1 << 20, 8), (5000, 3), (700, 7)])
def test_every_row_belongs_to_exactly_one_member(m, n, world):
    parts = [m.group_rows(n, world, r) for r in range(world)]
    allrows = np.concatenate(parts) if parts else np.zeros(0, dtype=np.uint32)
    assert allrows.size == n
    np.testing.assert_array_equal(np.sort(allrows), np.arange(n, dtype=np.uint32))
    for r, rows in enumerate(parts):
        assert np.all(np.diff(rows.astype(np.int64)) > 0)                # ascending
        assert np.all((rows // 256) % world == r)                         # row r' belongs to member (r' / 256) mod N
    sizes = [p.size for p in parts]
    assert max(sizes) - min(sizes) <= 256                                 # nobody gets more than one group extra


def test_two_members_interleave_in_groups_of_256(m):
    a, b = m.group_rows(1000, 2, 0), m.group_rows(1000, 2, 1)
    assert a[0] == 0 and a[255] == 255 and a[256] == 512
    assert b[0] == 256 and b[255] == 511 and b[256] == 768 and b[-1] == 999
    assert a.size == 512 and b.size == 488


def test_bad_arguments_are_refused(m):
    with pytest.raises(m.MsspeError):
        m.group_rows(10, 0, 0)
    with pytest.raises(m.MsspeError):
        m.group_rows(10, 2, 2)
    with pytest.raises(m.MsspeError):
        m.group_rows(-1, 2, 0)


def test_rccl_probe_reports_a_library_that_cannot_be_loaded(m):
    """group.hip's loader with the library name forced to a file that does not exist: a reason, no crash (the message
    used to be built from a second dlerror() call, which returns NULL) -- the branch transport "auto" takes when a
    multi-GPU node has no RCCL."""
    from msspe_amd.capi import rccl_available
    ok, why = rccl_available("/nonexistent/librccl-not-here.so.1")
    assert not ok
    assert "librccl-not-here" in why and "not loadable" in why
    ok2, why2 = rccl_available("/nonexistent/librccl-not-here.so.1")      # and again: dlerror state does not linger
    assert (ok2, why2) == (ok, why)
    # an existing library without the collective entry points is refused with its own reason
    ok3, why3 = rccl_available("libm.so.6")
    assert not ok3 and "entry point" in why3
