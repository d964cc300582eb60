#!/usr/bin/env python3
"""Development aid: long-oligo pools through the engine vs the CPU oracle (bit-exact dG / t)."""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT / "open-msspe-design_amd"), str(ROOT / "oracle")]
import numpy as np
import msspe_amd as m
import pyoracle


def main():
    ks = [int(x) for x in sys.argv[2:]] or [17, 18, 20, 21, 24, 28]
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 48
    eng = m.Engine(0)
    tables = pyoracle.Tables()
    bad = 0
    for k in ks:
        pool = m.synth.pool_strings(m.synth.random_pool(n, k, seed=900 + k))
        out = eng.cross_dimer(pool, m.Chem.ntthal(), -9000.0, want_dg=True, want_tm=True)
        stats = eng.pair_stage_stats()
        ovf = eng.last_overflow_pairs()
        cnt, dg, cf, tt = pyoracle.pool_pairs(tables, pool, pyoracle.ntthal_args(), -9000.0, want_t=True)
        ne = int((out["dg"] != dg).sum()) + int((out["tm"] != tt).sum())
        bits = np.unpackbits(out["bitmap"].view(np.uint8), axis=1, bitorder="little")[:, :n].astype(bool)
        nb = int((bits != cf.astype(bool)).sum())
        print(f"k={k} n={n}: mismatching doubles {ne}, decisions {nb}, handed on {ovf} of {n*n}, "
              f"reasons { {a: b for a, b in stats.items() if not isinstance(b, dict) and b} }", flush=True)
        if ne or nb:
            bad += 1
            w = np.argwhere(out["dg"] != dg)[:5]
            for i, j in w:
                print("   ", pool[i], pool[j], out["dg"][i, j], dg[i, j], out["tm"][i, j], tt[i, j])
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
