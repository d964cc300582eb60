#!/usr/bin/env python3
"""Quick GPU timing of the cross-dimer screen at a few pool sizes (development aid)."""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "open-msspe-design_amd"))
import numpy as np
import torch
import msspe_amd as m


def main():
    import os
    sizes = [int(x) for x in sys.argv[1:]] or [2048, 8192]
    K = int(os.environ.get("MSSPE_PROBE_K", "13"))
    if os.environ.get("MSSPE_PROBE_LIB"):
        from msspe_amd import capi
        capi.use_library(os.environ["MSSPE_PROBE_LIB"])
    eng = m.Engine(0)
    for kv in os.environ.get("MSSPE_PROBE_OPTIONS", "").split(","):
        if kv:
            eng.set_option(*kv.split("="))
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    chem = m.Chem.ntthal()
    for n in sizes:
        pool = m.pack_oligos(m.synth.random_pool(n, K))
        d_pool = torch.from_numpy(pool.view(np.int64)).cuda()
        words = (n + 63) // 64
        d_rc = torch.zeros(n, dtype=torch.int32, device="cuda")
        d_bm = torch.zeros((n, words), dtype=torch.int64, device="cuda")
        def run():
            d_rc.zero_()
            eng.cross_dimer_dev(d_pool.data_ptr(), n, K, chem, -9000.0, (0, n), (0, n),
                                d_rc.data_ptr(), d_bm.data_ptr())
        run()
        torch.cuda.synchronize()
        ovf = eng.last_overflow_pairs()
        stats = eng.pair_stage_stats()
        import os
        prof = os.environ.get('MSSPE_PROBE_PROFILE')
        if prof: eng.profile_enable(True)
        t0 = time.time()
        reps = 2
        for _ in range(reps):
            run()
        torch.cuda.synchronize()
        dt = (time.time() - t0) / reps
        if prof:
            nl, ms = eng.profile_read()
            print(f'   first-stage launches {nl}, avg {ms/max(nl,1):.3f} ms per launch', flush=True)
            eng.profile_enable(False)
        print(f"k={K} n={n}: {dt*1e3:.1f} ms/pass, {n*n/dt/1e6:.1f} M checks/s, conflicts={int(d_rc.sum())}, "
              f"overflow pairs={ovf} ({100.0*ovf/(n*n):.2f} %)", flush=True)
        print("   integer stage:", {k: f"{100.0*v/(n*n):.3f} %" for k, v in stats.items() if not isinstance(v, dict)},
              "| list mode:", {k: f"{100.0*v/(n*n):.3f} %" for k, v in stats["list"].items() if v}, flush=True)
        eng.pair_stage_stats()


if __name__ == "__main__":
    main()
