#!/usr/bin/env python3
"""Differential check of the long-oligo kernels (split-table integer stage + one-wave-per-pair f64
stage) against the dense f64 kernel on pools the CPU oracle would need hours for: identical conflict
bitmaps and counts on n^2 pairs, bit-identical dG / t on a sub-block.  Development aid (GPU)."""
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "open-msspe-design_amd"))
import numpy as np
import msspe_amd as m


def run(eng, pool, chem, dense_block):
    out = eng.cross_dimer(pool, chem, -9000.0, want_dg=False)
    sub = eng.cross_dimer(pool[:dense_block], chem, -9000.0, want_dg=True, want_tm=True)
    return out["bitmap"], out["row_conflicts"], sub["dg"], sub["tm"]


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    ks = [int(x) for x in sys.argv[2:]] or [15, 16, 17, 19, 21, 23, 25, 27, 29, 31, 32]
    eng = m.Engine(0)
    bad = 0
    for k in ks:
        for label, kw in (("default", {}), ("37C dv1.5 dntp0.6", dict(temp_c=37.0, dv=1.5, dntp=0.6))):
            pool = m.synth.pool_strings(m.synth.random_pool(n, k, seed=7000 + k))
            chem = m.Chem.ntthal(**kw)
            os.environ.pop("MSSPE_FORCE_GENERIC", None)
            fast = run(eng, pool, chem, 384)
            handed = eng.last_overflow_pairs()
            os.environ["MSSPE_FORCE_GENERIC"] = "1"
            slow = run(eng, pool, chem, 384)
            os.environ.pop("MSSPE_FORCE_GENERIC", None)
            eng.last_overflow_pairs()
            same = [bool(np.array_equal(a, b)) for a, b in zip(fast, slow)]
            print(f"k={k} {label}: n={n}, conflicts {int(fast[1].sum())}, handed on {handed}, "
                  f"bitmap/counts/dG/t equal: {same}", flush=True)
            bad += 0 if all(same) else 1
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
