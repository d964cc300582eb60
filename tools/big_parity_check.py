#!/usr/bin/env python3
"""One-off large GPU-vs-oracle comparison of the cross-dimer path (bit-exact dG / t matrices and
decisions) over several oligo lengths and chemistries; too slow on the CPU side for the test suite."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / 'open-msspe-design_amd')); sys.path.insert(0, str(ROOT / 'oracle'))
import numpy as np, time
import msspe_amd as m, pyoracle as o
eng=m.Engine(0); tabs=o.Tables()
cases = ((101,3000,13,{},-9000.0),(202,1500,12,{},-9000.0),(303,1200,15,{},-9000.0),(404,1000,16,{},-9000.0),
         (505,2000,10,{},-9000.0),(606,2000,13,dict(temp_c=37.0,mv=100.0,dv=1.5,dntp=0.2,dna_conc=50.0),-7000.0),
         (707,1500,14,dict(temp_c=60.0,dv=0.0),-3000.0))
for seed,n,k,kw,thr in cases:
    pool=m.synth.pool_strings(m.synth.random_pool(n,k,seed=seed))
    t0=time.time()
    out=eng.cross_dimer(pool,m.Chem.ntthal(**kw),thr,want_dg=True,want_tm=True)
    t1=time.time()
    cnt,dg,cf,tt=o.pool_pairs(tabs,pool,o.ntthal_args(**kw),thr,want_t=True)
    t2=time.time()
    ok_dg=np.array_equal(out["dg"],dg); ok_tm=np.array_equal(out["tm"],tt)
    bits=np.unpackbits(out["bitmap"].view(np.uint8),axis=1,bitorder="little")[:,:n].astype(bool)
    ok_cf=np.array_equal(bits,cf.astype(bool))
    print(f"k={k} n={n} {kw} thr={thr}: dg {ok_dg} tm {ok_tm} conflicts {ok_cf} ({int(cf.sum())})  gpu {t1-t0:.2f}s cpu {t2-t1:.1f}s stats {eng.pair_stage_stats()['needed_f64']}", flush=True)
    assert ok_dg and ok_tm and ok_cf
print("ALL OK")
