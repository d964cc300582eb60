#!/usr/bin/env python3
"""bench.py -- primer-pair thermo checks/sec (all-pairs cross-dimer) on N MI355X of one node.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one full screening pass of the hot path over one synthetic candidate pool that is
already resident in HBM: every ORDERED pair of the pool goes through the thermodynamic-alignment DP
(Primer3 thal ANY) down to the reference's conflict decision (od-msspe/src/delta_g.rs:61-153).
One check = one ordered pair.  N = 1: the 65,536-primer pool of BASELINE.md section 4
(4.29e9 checks).  N > 1: the pool grows with sqrt(N) so that every rank keeps 4.29e9 checks
("weak"); each rank owns n/N candidates, one RCCL all-gather assembles the packed pool, each rank
screens its row block against all columns, one RCCL all-reduce merges the per-primer conflict
counts.  Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT / "open-msspe-design_amd"))

import numpy as np
import torch
import torch.distributed as dist

import msspe_amd
from msspe_amd.distributed import all_reduce, gather_pool, reduce_counts, screen_row_block, shard_bounds

K = 13
THRESHOLD = -9000.0            # od-msspe/src/constants.rs:21
POOL_1GPU = 65536              # 1-GPU headline pool
FP64_PEAK_TFLOPS = 78.6        # MI355X FP64 vector peak = 1/2 of the 157.3 TF FP32 vector figure
HBM_PEAK_GBPS = 8000.0         # /opt/skills/guides/MI355X_MICROARCH.md:36 (spec)
# measured with the CPU oracle on 300^2 random 13-mer pairs (DESIGN.md "Algorithmic work"):
# double-precision add/mul/div/compare per check with the end terms evaluated once per cell
F64_OPS_PER_CHECK = 10458.0


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--pool", type=int, default=0, help="override the pool size (default 65536*sqrt(N))")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the cpu_baseline leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    return ap.parse_args()


def pool_size_for(n_gpus: int) -> int:
    n = POOL_1GPU * math.sqrt(n_gpus)
    q = 64 * n_gpus
    return int(round(n / q)) * q


def cpu_baseline(pool_ascii: np.ndarray, seconds: float, gpu_bitmap_rows: np.ndarray | None):
    """The CPU restatement of the reference path (the oracle), timed on this box's host cores on a
    bounded sample of the same workload: the first R rows x all columns.  Checker only: its
    decisions are compared with the GPU's for the same rows."""
    sys.path.insert(0, str(ROOT / "oracle"))
    import pyoracle
    tables = pyoracle.Tables()
    n = pool_ascii.shape[0]
    cores = min(os.cpu_count() or 1, 16)      # one GPU box grants about 16 host threads
    t0 = time.perf_counter()
    pyoracle.pool_pairs(tables, pool_ascii, rows=(0, 1), threads=1, want_dg=False, want_conflict=False)
    per_row_1t = max(time.perf_counter() - t0, 1e-6)
    rows = int(max(cores, min(n, seconds * cores * 0.6 / per_row_1t)))
    rows = max(1, min(rows, n))
    t0 = time.perf_counter()
    cnt, _, cf, _ = pyoracle.pool_pairs(tables, pool_ascii, rows=(0, rows), threads=cores,
                                        want_dg=False, want_conflict=True)
    dt = time.perf_counter() - t0
    agree = None
    if gpu_bitmap_rows is not None:
        got = np.unpackbits(gpu_bitmap_rows[:rows].view(np.uint8), axis=1, bitorder="little")[:, :n]
        agree = bool(np.array_equal(got.astype(bool), cf.astype(bool)))
    return {"value": rows * n / dt, "unit": "checks/s", "cores": cores, "kind": "port",
            "sample": f"rows 0..{rows - 1} x all {n} columns of the same pool ({rows * n} checks, "
                      f"{dt:.1f} s, OpenMP over rows); CPU restatement of the reference path, the "
                      f"reference binary (Rust + external ntthal) is not buildable offline",
            "single_thread_checks_per_s": n / per_row_1t,
            "decisions_equal_gpu": agree}


def main():
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # MSSPE_BENCH_BACKEND=gloo + MSSPE_BENCH_DEVICE=0 rehearses the N > 1 path with several ranks
    # on ONE card (tests only; RCCL refuses two ranks per device): never a reported number
    backend = os.environ.get("MSSPE_BENCH_BACKEND", "nccl")
    if backend != "nccl" and "MSSPE_BENCH_DEVICE" in os.environ:
        local_rank = int(os.environ["MSSPE_BENCH_DEVICE"])
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    n_gpus = max(args.gpus, world)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    n = args.pool if args.pool else pool_size_for(world)
    n = (n // (64 * world)) * (64 * world)
    shard = n // world
    r0, r1 = shard_bounds(n, world, rank)
    words = n // 64

    # synthetic pool (PCG64 seed 20260630); every rank uploads only the candidates it "produced"
    pool_ascii = msspe_amd.synth.random_pool(n, K)
    packed = msspe_amd.pack_oligos(pool_ascii)
    d_shard = torch.from_numpy(packed[r0:r1].view(np.int64).copy()).to(dev)
    d_conf = torch.zeros(n, dtype=torch.int32, device=dev)
    d_bitmap = torch.zeros((shard, words), dtype=torch.int64, device=dev)

    eng = msspe_amd.Engine(local_rank)
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    chem = msspe_amd.Chem.ntthal()

    def step():
        d_pool = gather_pool(d_shard, n)                       # RCCL all-gather over xGMI (N > 1)
        d_conf.zero_()
        screen_row_block(eng, d_pool, K, chem, THRESHOLD, (r0, r1), d_conf, d_bitmap)
        reduce_counts(d_conf)                                  # RCCL all-reduce of conflict counts
        return d_pool

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    eng.profile_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    launches, kernel_ms = eng.profile_read()
    eng.profile_enable(False)
    overflow = eng.last_overflow_pairs()
    stage = eng.pair_stage_stats()

    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    all_reduce(t, dist.ReduceOp.MAX)
    elapsed = float(t.item())
    checks_per_step = float(n) * float(n)
    value = checks_per_step * args.steps / elapsed

    if rank == 0:
        conflicts = int(d_conf.sum().item())
        # position-weighted sum of the merged per-primer counts: equal for every sharding of the same pool
        weights = torch.arange(1, n + 1, dtype=torch.int64, device=dev)
        conflict_checksum = int((d_conf.to(torch.int64) * weights).sum().item())
        # dominant kernel: the first-stage all-pairs kernel; one launch covers up to 2^27 checks of this rank's block
        checks_rank = float(shard) * float(n) * args.steps
        per_launch_checks = checks_rank / max(launches, 1)
        per_launch_s = kernel_ms / 1e3 / max(launches, 1)
        rows_per_launch = per_launch_checks / n
        # algorithmic HBM bytes of one launch: packed operands in, conflict bitmap + counts out
        bytes_per_launch = 8.0 * (rows_per_launch + n) + per_launch_checks / 8.0 + 4.0 * rows_per_launch
        kernel_checks_per_s = per_launch_checks / max(per_launch_s, 1e-12)
        traffic = None
        tf = ROOT / "profiles" / "traffic_latest.json"
        if tf.exists():
            try:
                traffic = json.loads(tf.read_text()).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        int_stage = os.environ.get("MSSPE_PAIR_KERNEL", "") != "f64"
        executed = None
        pf = ROOT / "profiles" / "pmc_latest.json"
        if pf.exists():
            try:
                executed = json.loads(pf.read_text())
            except Exception:
                executed = None
        roofline = {
            "bound": "valu",
            "note": ("vector-ALU bound DP: neither HBM nor MFMA binds it (SURVEY.md 8d). achieved = the "
                     "reference recurrence's f64 operations per check (oracle count) x checks / kernel time, "
                     "priced against the FP64 vector peak; the kernel itself runs the recurrence on exact "
                     "int32 (see executed) and replays the optimal path in f64. hbm gives algorithmic "
                     "bytes/s as the north star asks"),
            "kernel": "k_pairs_int" if int_stage else "k_pairs_fast",
            "achieved": kernel_checks_per_s * F64_OPS_PER_CHECK / 1e12,
            "peak": FP64_PEAK_TFLOPS,
            "unit": "TFLOP/s",
            "frac": kernel_checks_per_s * F64_OPS_PER_CHECK / 1e12 / FP64_PEAK_TFLOPS,
            "traffic": traffic,
            "f64_ops_per_check": F64_OPS_PER_CHECK,
            "launches": launches,
            "avg_launch_ms": per_launch_s * 1e3,
            "checks_per_launch": per_launch_checks,
            "executed": executed,
            "retried_in_list_mode": overflow / max(checks_rank * (args.steps + args.warmup) / args.steps, 1.0),
            "flagged_ties": stage["deferred"] / max(checks_rank * (args.steps + args.warmup) / args.steps, 1.0),
            "needed_f64_kernels": stage["needed_f64"] / max(checks_rank * (args.steps + args.warmup) / args.steps, 1.0),
            "hbm": {"achieved": bytes_per_launch / max(per_launch_s, 1e-12) / 1e9, "peak": HBM_PEAK_GBPS,
                    "unit": "GB/s",
                    "frac": bytes_per_launch / max(per_launch_s, 1e-12) / 1e9 / HBM_PEAK_GBPS,
                    "algorithmic_bytes_per_check": bytes_per_launch / per_launch_checks},
        }
        cpu = None
        if not args.no_cpu_baseline and world == 1:
            cpu = cpu_baseline(pool_ascii, args.cpu_seconds,
                               d_bitmap[:4096].cpu().numpy().view(np.uint64))
        out = {
            "metric": "primer-pair thermo checks/sec (all-pairs cross-dimer)",
            "value": value, "unit": "checks/s", "n_gpus": n_gpus, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "i32+f64", "data": "synthetic",
            "config": {"workload": f"cross-dimer all ordered pairs of {n} random 13-mers "
                                   f"({checks_per_step:.3g} checks/step), thal ANY at od-msspe defaults "
                                   f"(mv 50, dv 3, dNTP 0, 250 nM, 25 C), threshold -9000 cal/mol",
                       "pool": n, "kmer_size": K, "checks_per_step": checks_per_step,
                       "parallelism": f"row blocks x{world}" + (", all-gather pool + all-reduce counts "
                                                                 f"({'RCCL' if backend == 'nccl' else backend + ' REHEARSAL'})" if world > 1 else ""),
                       "conflicts": conflicts, "conflict_checksum": conflict_checksum,
                       "overflow_pairs_per_step": overflow / max(args.steps + args.warmup, 1)},
            "roofline": roofline,
            "cpu_baseline": cpu,
        }
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()
