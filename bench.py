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
counts.  `--config pool1m` is the strong-scaling form: BASELINE.json configs[3], the
1,048,576-candidate pool, cut into N row blocks (1.1e12 checks in total whatever N is).
Prints ONE JSON line on rank 0.

`--gpus N` without a launcher (WORLD_SIZE unset) starts N ranks itself with torch.distributed.run as a
child process, before this process touches the GPU, and exits with the child's status.
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
    ap.add_argument("--config", choices=["headline", "pool1m"], default="headline",
                    help="headline: 65,536 primers per GPU-equivalent (weak); pool1m: 1,048,576 primers in total (strong)")
    ap.add_argument("--stage-a", dest="stage_a", action="store_true", default=None,
                    help="also time stage A on 10,000 x 30 kb synthetic genomes (default: on at N = 1)")
    ap.add_argument("--no-stage-a", dest="stage_a", action="store_false")
    ap.add_argument("--stage-b", dest="stage_b", action="store_true", default=None,
                    help="also time stage B (Tm / GC / SELF_ANY / SELF_END / HAIRPIN) on 2,000 and 1,048,576 oligos "
                         "(default: on at N = 1)")
    ap.add_argument("--no-stage-b", dest="stage_b", action="store_false")
    ap.add_argument("--max-seconds", type=float, default=450.0,
                    help="--config pool1m only: budget of the WHOLE run after start-up (estimate pass + warm-up + timed "
                         "region + CPU leg); the step counts are cut to fit it, never below one timed step "
                         "(one step is 1.1e12 checks: about 8 minutes on one GPU, about a minute on eight)")
    ap.add_argument("--small-pool", dest="small_pool", action="store_true", default=None,
                    help="also time the reference-sized screens: 2,000 random 13-mers and the stage-A winners of the "
                         "10,000-genome alignment (default: on at N = 1)")
    ap.add_argument("--no-small-pool", dest="small_pool", action="store_false")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the cpu_baseline leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    return ap.parse_args()


def pool_size_for(n_gpus: int) -> int:
    n = POOL_1GPU * math.sqrt(n_gpus)
    q = 64 * n_gpus
    return int(round(n / q)) * q


def ntthal_live(pool_ascii: np.ndarray, pairs: int = 100000):
    """if the GPU box has Primer3's `ntthal` on $PATH, time the real thing (one
    process, one core, its -i stdin mode as od-msspe/src/delta_g.rs:93-145 drives it) on the first `pairs`
    ordered pairs of the pool.  None when there is no ntthal (the case on every box seen so far)."""
    import shutil
    import subprocess
    exe = shutil.which("ntthal")
    if exe is None:
        return None
    n = pool_ascii.shape[0]
    rows = max(1, min(n, pairs // n))
    prim = [bytes(r).decode() for r in pool_ascii]
    stdin = "\n".join(f"{prim[i]},{b}" for i in range(rows) for b in prim)
    t0 = time.perf_counter()
    res = subprocess.run([exe, "-a", "ANY", "-mv", "50.00", "-dv", "3.00", "-n", "0.00", "-d", "250.00", "-t", "25.00",
                          "-i"], input=stdin, capture_output=True, text=True)
    dt = time.perf_counter() - t0
    if res.returncode != 0:
        return {"error": res.stderr[-200:]}
    return {"value": rows * n / dt, "unit": "checks/s", "cores": 1, "kind": "reference",
            "sample": f"{rows * n} ordered pairs through {exe} -i, {dt:.1f} s"}


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
    live = ntthal_live(pool_ascii)
    return {"ntthal_live": live,
            "value": rows * n / dt, "unit": "checks/s", "cores": cores, "kind": "port",
            "sample": f"rows 0..{rows - 1} x all {n} columns of the same pool ({rows * n} checks, "
                      f"{dt:.1f} s, OpenMP over rows); CPU restatement of the reference path, the "
                      f"reference binary (Rust + external ntthal) is not buildable offline",
            "single_thread_checks_per_s": n / per_row_1t,
            "decisions_equal_gpu": agree}


def stage_a_line(eng, device):
    """Stage A (k-mer candidates, the HBM-bound part of the path) on BASELINE.json configs[2]'s shape:
    10,000 synthetic aligned genomes of 30 kb, both directions, the alignment resident in HBM in its packed
    form (2-bit bases + validity bit: what msspe_device_put_rows_packed leaves there).  Algorithmic bytes per
    direction: the windows' packed bits read (50 columns x 3 bits per segment), 38 keys x 4 B
    written, sorted with their 4-byte instance numbers and indexed once (post + word ids: 38 x 8 B), 38 x 4 B
    of count updates when the segment is covered (931 B per segment; rounds 1 and 2 priced 708 B: their JSON lines are
    not comparable in GB/s).  Returns (line, winners of the two directions)."""
    n_rows, length = 10000, 30000
    genomes = msspe_amd.synth.aligned_genomes(n_rows, length)
    d = eng.put_rows_packed(genomes)
    opt = msspe_amd.KmerOpt(500, 250, 50, K, 1000, 10)
    out = {}
    words = {0: [], 1: []}
    try:
        for direction in (0, 1):
            eng.kmer_candidates_packed(d, n_rows, length, opt, direction)
        torch.cuda.synchronize()
        reps = 3
        winners = 0
        its = np.zeros(4)
        ev_ms = []
        t0 = time.perf_counter()
        for _ in range(reps):
            for direction in (0, 1):
                # two events on the engine's stream around ONE direction: the device-side span of the call, the loop's
                # host round trips included (the call returns with the winners on the host)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                w, _f = eng.kmer_candidates_packed(d, n_rows, length, opt, direction)
                e1.record()
                ev_ms.append((e0, e1))
                winners += len(w)
                words[direction] = w
                its += [eng.info("stage_a_" + x) for x in ("fast_iterations", "general_iterations", "rebuilds", "idle_iterations")]
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / (2 * reps) * 1e3
        ev = float(np.mean([a.elapsed_time(b) for a, b in ev_ms]))
        # both directions in ONE call (msspe_kmer_candidates_both_packed_dev: the second direction on a second stream
        # from a second host thread), as the C++ pipeline runs stage A
        eng.kmer_candidates_both_packed(d, n_rows, length, opt)
        t1 = time.perf_counter()
        for _ in range(reps):
            both = eng.kmer_candidates_both_packed(d, n_rows, length, opt)
        both_ms = (time.perf_counter() - t1) / reps * 1e3
        both_equal = all(list(both[x][0]) == list(words[x]) for x in (0, 1))
    finally:
        eng.device_free(d)
    segments = n_rows * ((length - 500) // 250 + 1)
    alg_bytes = segments * (19.0 + 38 * 4.0 + 38 * 8.0 + 38 * 8.0 + 38 * 4.0)
    its /= 2 * reps
    out.update({"workload": f"{n_rows} synthetic aligned genomes x {length} columns (packed), k = {K}, both directions",
                "ms_per_direction": ms, "winners_per_direction": winners / (2 * reps),
                "greedy_iterations_per_direction": {"from_partition_leaders": its[0], "after_posting_walks": its[1],
                                                    "candidate_lists_made": its[2], "idle": its[3]},
                "algorithmic_bytes_per_direction": alg_bytes,
                "algorithmic_bytes_per_segment": alg_bytes / segments,
                "both_directions_one_call_ms": both_ms, "both_directions_one_call_equal": bool(both_equal),
                "GBps": alg_bytes / (ms * 1e-3) / 1e9,
                "frac_of_6.29TBps_copy_ceiling": alg_bytes / (ms * 1e-3) / 6.29e12,
                "note": "ms_per_direction: host wall time per direction incl. the greedy loop's dependent launches and "
                        "its host round trips"})
    # measured in THIS run: algorithmic bytes / the events' span of one direction, against the measured-copy ceiling
    out["roofline"] = {"bound": "hbm", "achieved": alg_bytes / (ev * 1e-3) / 1e9, "peak": 6290.0, "unit": "GB/s",
                       "frac": alg_bytes / (ev * 1e-3) / 6.29e12, "event_ms_per_direction": ev,
                       "source": "live: HIP events on the engine's stream around each direction, mean of %d" % len(ev_ms)}
    roof = stage_a_kernel_roofline(alg_bytes)
    if roof:
        out["roofline_sum_of_kernels"] = roof
    return out, words


def stage_a_kernel_roofline(alg_bytes: float):
    """bytes / SUM of kernel time per direction, from the newest committed rocprofv3 summary of the same workload
    (tools/perf_stage_a.py 10000 30000 under --kernel-trace --stats: two directions x two repetitions; the packing of
    byte rows, which the packed entry point does not run, and the engine's one-off LDS probe are left out).  IMPORTED,
    and labelled so -- bench.py cannot see kernel times of this many small launches without the profiler; the figure
    measured in this run is stage_a.roofline.  The summary's side file names the commit it was taken at."""
    import csv
    prof = Path(__file__).resolve().parent / "profiles"
    cands = sorted(prof.glob("r*_stage_a_kernel_stats.csv"))
    if not cands:
        return None
    path = cands[-1]
    total_ns = 0.0
    for row in csv.DictReader(open(path)):
        name = row["Name"]
        if "k_pack_rows" in name or "k_lds_" in name:
            continue
        total_ns += float(row["TotalDurationNs"])
    ms = total_ns / 4 / 1e6
    commit = None
    side = path.with_suffix(".commit")
    if side.exists():
        commit = side.read_text().strip()
    return {"bound": "hbm", "achieved": alg_bytes / (ms * 1e-3) / 1e9, "peak": 6290.0, "unit": "GB/s",
            "frac": alg_bytes / (ms * 1e-3) / 6.29e12, "kernel_ms_per_direction": ms, "profiled_commit": commit,
            "source": f"profiles/{path.name} (imported: rocprofv3 kernel times of the same workload, not this run)"}


def small_pool_line(eng, device, winners):
    """The screens the reference really runs: od-msspe caps candidates at 1,000 per direction (main.rs:344), so its
    stage-C matrix is at most 2,000^2.  Two pools, resident in HBM, counts + bitmap: 2,000 random 13-mers (SURVEY.md
    8d) and the actual stage-A winners of the 10,000-genome alignment (both directions).  ms per FULL screen from HIP
    events on the engine's stream (composition sort, first stage, every list stage and the flush inside), the CPU
    restatement on the same pool beside it, every decision compared."""
    sys.path.insert(0, str(ROOT / "oracle"))
    import pyoracle
    tables = pyoracle.Tables()
    chem = msspe_amd.Chem.ntthal()
    pools = {"random_2000": msspe_amd.synth.random_pool(2000, K)}
    if winners and (winners[0] or winners[1]):
        w = list(winners[0]) + list(winners[1])
        pools["stage_a_winners"] = np.frombuffer("".join(w).encode(), dtype=np.uint8).reshape(len(w), K)
    out = {"chemistry": "od-msspe defaults (mv 50, dv 3, dNTP 0, 250 nM, 25 C), threshold -9000 cal/mol", "pools": {}}
    cores = min(os.cpu_count() or 1, 16)
    for name, pool_ascii in pools.items():
        n = pool_ascii.shape[0]
        words = (n + 63) // 64
        d_pool = torch.from_numpy(msspe_amd.pack_oligos(pool_ascii).view(np.int64)).to(device)
        d_rc = torch.zeros(n, dtype=torch.int32, device=device)
        d_bm = torch.zeros((n, words), dtype=torch.int64, device=device)

        def screen():
            d_rc.zero_()
            eng.cross_dimer_dev(d_pool.data_ptr(), n, K, chem, THRESHOLD, (0, n), (0, n), d_rc.data_ptr(), d_bm.data_ptr())
        for _ in range(3):
            screen()
        torch.cuda.synchronize()
        eng.last_overflow_pairs()
        eng.profile_enable(True)
        reps = 30
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            screen()
        e1.record()
        torch.cuda.synchronize()
        launches, kernel_ms = eng.profile_read()
        eng.profile_enable(False)
        handed = eng.last_overflow_pairs() / reps
        ms = e0.elapsed_time(e1) / reps
        t0 = time.perf_counter()
        _cnt, _, cf, _ = pyoracle.pool_pairs(tables, pool_ascii, threads=cores, want_dg=False, want_conflict=True)
        dt = time.perf_counter() - t0
        got = np.unpackbits(d_bm.cpu().numpy().view(np.uint8), axis=1, bitorder="little")[:, :n]
        out["pools"][name] = {
            "primers": n, "checks": n * n, "ms_per_screen": ms, "checks_per_s": n * n / (ms * 1e-3),
            "first_stage_launches_per_screen": launches / reps, "first_stage_ms": kernel_ms / max(launches, 1),
            "handed_to_list_stages": handed / (n * n), "conflicts": int(d_rc.sum().item()),
            "cpu_baseline": {"value": n * n / dt, "unit": "checks/s", "cores": cores, "kind": "port",
                             "sample": f"the whole pool, {n * n} checks, {dt:.2f} s, OpenMP over rows",
                             "decisions_equal_gpu": bool(np.array_equal(got.astype(bool), cf.astype(bool)))}}
        del d_pool, d_rc, d_bm
    stats = ROOT / "profiles" / "r04_small_pool_kernel_stats.csv"
    if stats.exists():
        out["kernel_launches"] = f"profiles/{stats.name} (rocprofv3 --kernel-trace --stats of tools/perf_small_pool.py 2000)"
    return out


def stage_b_line(eng, device, cpu_seconds: float):
    """Stage B (od-msspe/src/primer.rs:143-166: what the reference gets from primer3_core per candidate) with the
    oligos resident in HBM: Tm + GC %, SELF_ANY, SELF_END (thal ANY / END1 of the oligo with itself) and HAIRPIN,
    at 2,000 oligos (what the reference's loop produces, main.rs:344) and at 1,048,576 (configs[3]'s pool).
    Every part is timed on its own with HIP events on the engine's stream; the CPU restatement runs beside it on a
    bounded sample and its five numbers per oligo are compared bit for bit."""
    chem = msspe_amd.Chem.primer3()
    out = {"chemistry": "primer3_core defaults (mv 50, dv 1.5, dNTP 0.6, 50 nM, 37 C)", "sizes": {}}
    for n in (2000, 1 << 20):
        pool_ascii = msspe_amd.synth.random_pool(n, K, seed=2000 + n)
        d_pool = torch.from_numpy(msspe_amd.pack_oligos(pool_ascii).view(np.int64)).to(device)
        d_out = torch.zeros((5, n), dtype=torch.float64, device=device)
        ptr = [d_out[q].data_ptr() for q in range(5)]
        # self_any / self_end alone: one DP fill each; self_dimers: both from ONE fill (END1 is the same fillMatrix with
        # the terminal pick restricted to the last row), which is what "all" and the pipeline run
        parts = {"tm_gc": (ptr[0], ptr[1], 0, 0, 0), "self_any": (0, 0, ptr[2], 0, 0),
                 "self_end": (0, 0, 0, ptr[3], 0), "self_dimers": (0, 0, ptr[2], ptr[3], 0),
                 "hairpin": (0, 0, 0, 0, ptr[4]), "all": tuple(ptr)}
        times = {}
        for name, a in parts.items():
            call = lambda: eng.oligo_stats_dev(d_pool.data_ptr(), n, K, chem, *a)
            call()
            torch.cuda.synchronize()
            reps = 20 if n <= 4096 else 3
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                call()
            e1.record()
            torch.cuda.synchronize()
            times[name] = e0.elapsed_time(e1) / reps
        entry = {"ms": times, "oligos_per_s": n / (times["all"] * 1e-3),
                 "dp_checks_per_s": {q: n / (times[q] * 1e-3) for q in ("self_any", "self_end", "hairpin")},
                 # the two dimer DPs of an oligo with itself are ordinary thal checks: priced like the headline
                 # kernel, reference f64 operations per check against the FP64 vector peak
                 "roofline_frac_self_dimers": 2 * n * F64_OPS_PER_CHECK /
                                              (times["self_dimers"] * 1e-3) / 1e12 / FP64_PEAK_TFLOPS}
        out["sizes"][str(n)] = entry
        if n > 2000:
            # the CPU restatement beside it, on a bounded sample of the same oligos (OpenMP over oligos)
            sys.path.insert(0, str(ROOT / "oracle"))
            import pyoracle
            tables = pyoracle.Tables()
            m_cpu = 1 << 18
            words = msspe_amd.synth.pool_strings(pool_ascii[:m_cpu])
            cores = min(os.cpu_count() or 1, 16)
            t0 = time.perf_counter()
            ref = pyoracle.check_primers(tables, words)
            dt = time.perf_counter() - t0
            got = d_out[:, :m_cpu].cpu().numpy()
            equal = all(np.array_equal(got[q], ref[key]) for q, key in
                        enumerate(("tm", "gc", "self_any_th", "self_end_th", "hairpin_th")))
            out["cpu_baseline"] = {"value": m_cpu / dt, "unit": "oligos/s", "cores": cores, "kind": "port",
                                   "sample": f"the first {m_cpu} of the {n} oligos, {dt:.2f} s, OpenMP over oligos; CPU "
                                             f"restatement of primer3_core's check_primers arithmetic",
                                   "five_statistics_equal_gpu": bool(equal)}
        del d_pool, d_out
    return out


def respawn_under_launcher(args):
    """`python bench.py --gpus N` with no launcher: start N ranks as a CHILD process (never exec from a
    process that may have touched the GPU; this one has not) and hand its exit status on."""
    import subprocess
    port = 29500 + os.getpid() % 400
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:]
    return subprocess.run(cmd).returncode


def main():
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(respawn_under_launcher(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE is {world}: launch one rank per GPU "
                 f"(python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus})")
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
    n_gpus = world
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    strong = args.config == "pool1m"
    if strong and rank == 0:
        # a step of the 1,048,576 pool is minutes of queued launches behind one fence: say on stderr that the run is
        # alive (a watchdog that sees no output for minutes takes a run for hung); stdout stays the one JSON line
        import threading
        t_alive = time.perf_counter()

        def heartbeat():
            while True:
                time.sleep(60.0)
                print(f"bench.py pool1m: running, {time.perf_counter() - t_alive:.0f} s", file=sys.stderr, flush=True)
        threading.Thread(target=heartbeat, daemon=True).start()
    n = args.pool if args.pool else (1 << 20 if strong else pool_size_for(world))
    if not args.pool:
        n = (n // (64 * world)) * (64 * world)
    r0, r1 = shard_bounds(n, world, rank)       # row blocks may differ by one row (any n, any N)
    shard = r1 - r0
    words = (n + 63) // 64

    # synthetic pool (PCG64 seed 20260630); every rank uploads only the candidates it "produced"
    pool_ascii = msspe_amd.synth.random_pool(n, K)
    packed = msspe_amd.pack_oligos(pool_ascii)
    d_shard = torch.from_numpy(packed[r0:r1].view(np.int64).copy()).to(dev)
    d_conf = torch.zeros(n, dtype=torch.int32, device=dev)
    # the conflict bitmap of a rank's row block: 8 B per 64 columns; at 1M candidates that is 137 GB for a
    # single rank, so the largest pools run with counts only (SURVEY.md 7 "output volume")
    want_bitmap = float(shard) * words * 8 <= 64e9
    d_bitmap = torch.zeros((shard, words), dtype=torch.int64, device=dev) if want_bitmap else None

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

    steps, warmup = args.steps, args.warmup
    t_start = time.perf_counter()
    if strong:
        # One step of the 1,048,576 pool is 1.1e12 checks (minutes).  The step counts are cut so that the WHOLE run --
        # this estimate, the warm-up, the timed region and the CPU leg -- fits --max-seconds (450 s: the driver's limit
        # is 600 s), but never below one timed step (on one GPU a single step is about 8 minutes: no budget holds it).
        # The estimate: 1/16 of the rank's rows against all columns, fenced (it also builds the tables and sizes the
        # lists: the first warm-up).  Every rank takes the slowest rank's estimate, so all run the same counts.
        fence()
        t0 = time.perf_counter()
        d_pool0 = gather_pool(d_shard, n)
        est_rows = max(1, shard // 16)
        d_conf.zero_()
        screen_row_block(eng, d_pool0, K, chem, THRESHOLD, (r0, r0 + est_rows), d_conf,
                         d_bitmap[:est_rows] if d_bitmap is not None else None)
        fence()
        est = torch.tensor([(time.perf_counter() - t0) * shard / est_rows * 1.03], dtype=torch.float64, device=dev)
        all_reduce(est, dist.ReduceOp.MAX)
        per_step = max(float(est.item()), 1e-3)
        left = args.max_seconds - (time.perf_counter() - t_start) - (0.0 if args.no_cpu_baseline else args.cpu_seconds + 5.0)
        steps = max(1, min(args.steps, int(left / per_step)))
        warmup = max(0, min(args.warmup, int((left - steps * per_step) / per_step)))
        warmup_done = 0
        del d_pool0
        eng.last_overflow_pairs()      # the estimate's hand-overs and stage counters are not the steps'
        eng.pair_stage_stats()
    else:
        warmup_done = 0
    for _ in range(warmup):
        step()
    warmup_done += warmup
    fence()
    eng.profile_enable(True)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    steps_requested, warmup_requested = args.steps, args.warmup
    args.steps, args.warmup = steps, warmup_done     # what ran: every figure below is per step that ran
    launches, kernel_ms = eng.profile_read()
    eng.profile_enable(False)
    overflow = eng.last_overflow_pairs()
    stage = eng.pair_stage_stats()
    # per-rank time in the dominant kernel (HIP events on the engine's stream): a load imbalance between
    # the row blocks would show as a spread here
    km = torch.tensor([kernel_ms], dtype=torch.float64, device=dev)
    km_lo, km_hi = km.clone(), km.clone()
    all_reduce(km_lo, dist.ReduceOp.MIN)
    all_reduce(km_hi, dist.ReduceOp.MAX)

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
        # dominant kernel: the first-stage all-pairs kernel; one launch covers up to 2^29 checks of this rank's block
        checks_rank = float(shard) * float(n) * args.steps
        per_launch_checks = checks_rank / max(launches, 1)
        per_launch_s = kernel_ms / 1e3 / max(launches, 1)
        rows_per_launch = per_launch_checks / n
        # algorithmic HBM bytes of one launch: packed operands in, conflict bitmap + counts out
        bytes_per_launch = 8.0 * (rows_per_launch + n) + per_launch_checks / 8.0 + 4.0 * rows_per_launch
        kernel_checks_per_s = per_launch_checks / max(per_launch_s, 1e-12)
        # HBM traffic from the PMC passes (profiles/traffic_latest.json: FETCH_SIZE / WRITE_SIZE of the same kernel on
        # this bench's pool, per check) scaled to THIS run's launch, so that it sits beside checks_per_launch /
        # avg_launch_ms / algorithmic_bytes in the same unit: bytes per launch
        traffic = None
        traffic_per_check = None
        tf = ROOT / "profiles" / "traffic_latest.json"
        if tf.exists():
            try:
                tj0 = json.loads(tf.read_text())
                traffic_per_check = float(tj0["hbm_bytes_per_launch"]) / float(tj0["checks_per_launch"])
                traffic = traffic_per_check * per_launch_checks
            except Exception:
                traffic = None
        # counters come from a separate rocprofv3 --pmc run (profiles/): they are IMPORTED, not measured
        # in this run, and say so; only launches / avg_launch_ms / achieved / frac below are live
        executed = None
        pf = ROOT / "profiles" / "pmc_latest.json"
        if pf.exists():
            try:
                executed = json.loads(pf.read_text())
                executed["imported_from"] = "profiles/pmc_latest.json"
            except Exception:
                executed = None
        traffic_src = None
        if traffic is not None:
            try:
                tj = json.loads(tf.read_text())
                traffic_src = {"imported_from": "profiles/traffic_latest.json", "kernel": tj.get("kernel"),
                               "commit": tj.get("commit"), "round": tj.get("round")}
            except Exception:
                traffic_src = {"imported_from": "profiles/traffic_latest.json"}
        roofline = {
            "bound": "valu",
            "note": ("vector-ALU bound DP: neither HBM nor MFMA binds it (SURVEY.md 8d). achieved = the "
                     "reference recurrence's f64 operations per check (oracle count) x checks / kernel time, "
                     "priced against the FP64 vector peak; the kernel itself runs the recurrence on exact "
                     "int32 (see executed) and replays the optimal path in f64. hbm gives algorithmic "
                     "bytes/s as the north star asks"),
            "kernel": "k_pairs_row",
            "achieved": kernel_checks_per_s * F64_OPS_PER_CHECK / 1e12,
            "peak": FP64_PEAK_TFLOPS,
            "unit": "TFLOP/s",
            "frac": kernel_checks_per_s * F64_OPS_PER_CHECK / 1e12 / FP64_PEAK_TFLOPS,
            "traffic": traffic,
            "traffic_unit": "HBM bytes per launch of checks_per_launch checks (imported PMC bytes per check x this run's launch)",
            "traffic_bytes_per_check": traffic_per_check,
            "algorithmic_bytes": bytes_per_launch,
            "traffic_source": traffic_src,
            "f64_ops_per_check": F64_OPS_PER_CHECK,
            "launches": launches,
            "avg_launch_ms": per_launch_s * 1e3,
            "checks_per_launch": per_launch_checks,
            "kernel_ms_per_rank": {"min": float(km_lo.item()) / args.steps, "max": float(km_hi.item()) / args.steps,
                                   "note": "dominant kernel, per step, over the ranks"},
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
        if not args.no_cpu_baseline:      # rank 0, every N: rows 0.. belong to rank 0's block
            cpu = cpu_baseline(pool_ascii, args.cpu_seconds,
                               d_bitmap[:min(4096, shard)].cpu().numpy().view(np.uint64) if want_bitmap else None)
        stage_a = None
        winners = None
        if args.stage_a if args.stage_a is not None else (world == 1 and not args.pool and not strong):
            stage_a, winners = stage_a_line(eng, dev)
        small_pool = None
        if args.small_pool if args.small_pool is not None else (world == 1 and not args.pool and not strong):
            small_pool = small_pool_line(eng, dev, winners)
        stage_b = None
        if args.stage_b if args.stage_b is not None else (world == 1 and not args.pool and not strong):
            stage_b = stage_b_line(eng, dev, args.cpu_seconds)
        out = {
            "metric": "primer-pair thermo checks/sec (all-pairs cross-dimer)",
            "value": value, "unit": "checks/s", "n_gpus": n_gpus, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
            "dtype": "i32+f64", "data": "synthetic",
            "config": {"workload": ("BASELINE configs[3], strong scaling: " if strong else "") +
                                   f"cross-dimer all ordered pairs of {n} random 13-mers "
                                   f"({checks_per_step:.3g} checks/step), thal ANY at od-msspe defaults "
                                   f"(mv 50, dv 3, dNTP 0, 250 nM, 25 C), threshold -9000 cal/mol",
                       "pool": n, "kmer_size": K, "checks_per_step": checks_per_step,
                       "parallelism": f"row blocks x{world}" + (", all-gather pool + all-reduce counts "
                                                                 f"({'RCCL' if backend == 'nccl' else backend + ' REHEARSAL'})" if world > 1 else ""),
                       "conflicts": conflicts, "conflict_checksum": conflict_checksum,
                       "overflow_pairs_per_step": overflow / max(args.steps + args.warmup, 1)},
            "roofline": roofline,
            "cpu_baseline": cpu,
            "stage_a": stage_a,
            "stage_b": stage_b,
            "small_pool": small_pool,
        }
        out["timed_region_s"] = elapsed
        if strong:
            out["steps_requested"], out["warmup_requested"] = steps_requested, warmup_requested
            out["note"] = (f"pool1m: step counts cut to the --max-seconds {args.max_seconds:.0f} budget of the whole run "
                           f"({per_step:.1f} s per step estimated from 1/16 of the rows); steps / warmup are what ran")
        print(json.dumps(out))
    if world > 1:
        dist.barrier()   # rank 0 is still timing the CPU baseline: the ranks leave together
        dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()
