#!/usr/bin/env python3
"""GPU timing of the reference-realistic cross-dimer screens (development aid; bench.py's `small_pool` reports the same):
N = 2,000 random 13-mers (od-msspe caps candidates at 1,000 per direction, main.rs:344) and the actual stage-A winners
of an alignment, pool resident, counts + bitmap, HIP events on the engine's stream.
usage: perf_small_pool.py [n ...] [--reps R] [--edges]"""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "open-msspe-design_amd"))
import numpy as np
import torch
import msspe_amd as m


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    reps = int(sys.argv[sys.argv.index("--reps") + 1]) if "--reps" in sys.argv else 50
    if "--reps" in sys.argv:
        args.remove(sys.argv[sys.argv.index("--reps") + 1])
    sizes = [int(x) for x in args] or [2000]
    K = 13
    eng = m.Engine(0)
    import os
    for kv in os.environ.get("MSSPE_PROBE_OPTIONS", "").split(","):
        if kv:
            eng.set_option(*kv.split("="))
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    chem = m.Chem.ntthal()
    for n in sizes:
        d_pool = torch.from_numpy(m.pack_oligos(m.synth.random_pool(n, K)).view(np.int64)).cuda()
        words = (n + 63) // 64
        d_rc = torch.zeros(n, dtype=torch.int32, device="cuda")
        d_bm = torch.zeros((n, words), dtype=torch.int64, device="cuda")
        d_edges = torch.zeros((1 << 20, 2), dtype=torch.int64, device="cuda")
        d_cnt = torch.zeros(1, dtype=torch.int64, device="cuda")

        def bitmap():
            d_rc.zero_()
            eng.cross_dimer_dev(d_pool.data_ptr(), n, K, chem, -9000.0, (0, n), (0, n), d_rc.data_ptr(), d_bm.data_ptr())

        def edges():
            eng.cross_dimer_edges_dev(d_pool.data_ptr(), n, K, chem, -9000.0, (0, n), (0, n), d_edges.data_ptr(), 1 << 20,
                                      d_cnt.data_ptr())

        for name, fn in (("counts+bitmap", bitmap), ("edge list", edges)):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            eng.profile_enable(True)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            nl, kms = eng.profile_read()
            eng.profile_enable(False)
            ms = e0.elapsed_time(e1) / reps
            ovf = eng.last_overflow_pairs()
            print(f"n={n} {name}: {ms:.3f} ms per screen, handed on {ovf / (reps + 3) / (n * n) * 100:.2f} %, {n * n / ms / 1e6:.1f} G checks/s x1e-3, first stage "
                  f"{kms / max(nl, 1):.3f} ms x {nl // reps} launch(es); conflicts {int(d_rc.sum())} edges {int(d_cnt.item())}",
                  flush=True)
    eng.close()


if __name__ == "__main__":
    main()
