"""The od-msspe pipeline on N GPUs of one node, one process per GPU (BASELINE.json configs[4]).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        -m msspe_amd.pipeline_ranks -i aligned.fasta -o primers.csv [--kmer-size 13 --max-iterations 1000 ...]

What shards and what does not:
    stage A   the greedy max-cover loop is sequential and its working set fits one device: rank 0 runs it
              (both directions), the winners are broadcast (replicas only, no data-path collective)
    stage B   oligos are independent: rank r computes Tm / GC / SELF_ANY / SELF_END / HAIRPIN for its slice
              of the candidates, one all-gather assembles the five arrays everywhere; mean / sigma and the
              filter (od-msspe/src/main.rs:408-516) are then the same cheap host arithmetic on every rank
    stage C   every ordered pair belongs to exactly one row block: rank r screens rows [r0, r1) of the pool
              against all columns and emits its conflict edges (msspe_cross_dimer_edges_dev); the edge lists
              are gathered on rank 0, which runs the vertex cover (main.rs:754-815) and writes the CSV
The filter / statistics / vertex cover / CSV text come from the C++ host layer (libod_msspe_host.so), the
same code the single-process CLI `od-msspe-hip` runs, so the two produce the same file.
The input must be aligned already (the single-process CLI spawns MAFFT; a rank that holds a GPU does not).
"""
from __future__ import annotations

import argparse
import ctypes as C
import os
import sys
from pathlib import Path

import numpy as np
import torch
import torch.distributed as dist

from . import capi
from .distributed import all_reduce, dealt_rows, screen_dealt_rows_edges, shard_bounds

PKG = Path(__file__).resolve().parent.parent


def _host():
    capi.load_library()
    return C.CDLL(str(PKG / "libod_msspe_host.so"))


def _call(fn, *args):
    """A host-layer text call: (status, text).  The output buffer is sized from the inputs (none of these calls
    returns more than it was given, plus formatting) and grown while the library reports that it was too small
    (c_hooks.cpp emit(): status -3)."""
    cap = max(1 << 16, 2 * sum(len(x) for x in args if isinstance(x, (bytes, bytearray))) + (1 << 12))
    while True:
        buf = C.create_string_buffer(cap)
        rc = fn(*args, buf, cap)
        if rc != -3 or cap >= (1 << 34):
            return rc, buf.value.decode()
        cap *= 4


def parse_args(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("-i", "--input", required=True)
    ap.add_argument("-o", "--output", required=True)
    # names and defaults of od-msspe/src/config.rs:11-148 and constants.rs:1-21
    ap.add_argument("--kmer-size", type=int, default=13)
    ap.add_argument("--window-size", type=int, default=500)
    ap.add_argument("--overlap-size", type=int, default=250)
    ap.add_argument("--search-windows-size", type=int, default=50)
    ap.add_argument("--max-iterations", type=int, default=1000)
    ap.add_argument("--max-mismatch-segments", type=int, default=-1)
    ap.add_argument("--min-tm", type=float, default=30.0)
    ap.add_argument("--max-tm", type=float, default=60.0)
    ap.add_argument("--tm-stddev", type=float, default=2.0)
    ap.add_argument("--max-self-dimer-any-tm", type=float, default=47.0)
    ap.add_argument("--max-self-dimer-end-tm", type=float, default=47.0)
    ap.add_argument("--max-hairpin-tm", type=float, default=24.0)
    ap.add_argument("--delta-g-threshold", type=float, default=-9000.0)
    ap.add_argument("--mv-conc", type=float, default=50.0)
    ap.add_argument("--dv-conc", type=float, default=3.0)
    ap.add_argument("--dntp-conc", type=float, default=0.0)
    ap.add_argument("--dna-conc", type=float, default=250.0)
    ap.add_argument("--annealing-temp", type=float, default=25.0)
    for flag in ("check-cross-dimers", "check-self-dimers", "check-hairpin"):
        ap.add_argument("--" + flag, choices=["true", "false"], default="true")
    for flag in ("keep-all", "disable-tm-stddev", "disable-min-max-tm"):
        ap.add_argument("--" + flag, choices=["true", "false"], default="false")
    return ap.parse_args(argv)


def _two(v: float) -> float:
    """ntthal receives "{:.2}" strings of the f32 options (od-msspe/src/delta_g.rs:98-106)."""
    return float("%.2f" % float(np.float32(v)))


def main(argv=None) -> int:
    a = parse_args(argv)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("MSSPE_BENCH_BACKEND", "nccl")     # gloo + MSSPE_BENCH_DEVICE: rehearsal on one card
    if backend != "nccl" and "MSSPE_BENCH_DEVICE" in os.environ:
        local_rank = int(os.environ["MSSPE_BENCH_DEVICE"])
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    host = _host()
    eng = capi.Engine(local_rank)
    k = a.kmer_size

    # ---- stage A on rank 0, winners to everybody --------------------------------------------------
    payload = [None]
    if rank == 0:
        rc, rec_text = _call(host.odm_to_records, Path(a.input).read_bytes())
        if rc == -3:
            raise SystemExit("pipeline_ranks: record buffer too small for the input")
        if rc < 0 or not rec_text:
            raise SystemExit("No sequences found in the input file")
        records = [ln.split("\t") for ln in rec_text.splitlines()]
        length = max(len(s) for _n, s in records)
        arr = np.full((len(records), length), ord("-"), dtype=np.uint8)
        for i, (_n, s) in enumerate(records):
            arr[i, : len(s)] = np.frombuffer(s.encode(), dtype=np.uint8)
        mm = a.max_mismatch_segments if a.max_mismatch_segments >= 0 else min(10, max(1, -(-len(records) // 50)))
        opt = capi.KmerOpt(a.window_size, a.overlap_size, a.search_windows_size, k, a.max_iterations, mm)
        d_aln = torch.from_numpy(arr).to(dev)
        cand = []
        for direction in (0, 1):
            words, _f = eng.kmer_candidates(None, opt, direction, device_ptr=d_aln.data_ptr(), n_seq=arr.shape[0],
                                            seq_len=length)
            cand.append(words)
        del d_aln
        payload = [cand]
    if world > 1:
        dist.broadcast_object_list(payload, src=0)
    cand_f, cand_r = payload[0]

    # ---- stage B: slices of the candidates, one all-gather ----------------------------------------
    def stats_of(words):
        n = len(words)
        out = np.zeros((5, n))
        if n:
            r0, r1 = shard_bounds(n, world, rank)
            mine = np.zeros((5, n))
            if r1 > r0:
                st = eng.oligo_stats(words[r0:r1])
                for q, key in enumerate(("tm", "gc", "self_any", "self_end", "hairpin")):
                    mine[q, r0:r1] = st[key]
            if world > 1:
                # disjoint slices: the sum is the concatenation.  Reduced on the device: the RCCL process group
                # has no CPU backend (a host tensor would raise "No backend type associated with device type cpu");
                # under gloo the helper stages the device tensor through the host
                t = torch.from_numpy(mine).to(dev)
                all_reduce(t)
                mine = t.cpu().numpy()
            out = mine
        rnd = capi.round_fixed_f32     # the text primer3_core prints, read back as f32 (primer.rs:94-106)
        return {"tm": np.array([rnd(x, 3) for x in out[0]], dtype=np.float32),
                "gc": np.array([rnd(x, 3) for x in out[1]], dtype=np.float32),
                "any": np.array([rnd(x, 2) for x in out[2]], dtype=np.float32),
                "end": np.array([rnd(x, 2) for x in out[3]], dtype=np.float32),
                "hp": np.array([rnd(x, 2) for x in out[4]], dtype=np.float32)}

    def kmer_stats(words, direction):
        s = stats_of(words)
        n = len(words)
        rows = []
        if not n:
            return rows
        tm = np.ascontiguousarray(s["tm"])
        std = C.c_float()
        host.odm_tm_stat.restype = C.c_float
        mean = host.odm_tm_stat(tm.ctypes.data_as(C.POINTER(C.c_float)), n, 0, C.byref(std))
        for i, w in enumerate(words):
            tm_ok = abs(np.float32(tm[i]) - np.float32(mean)) <= np.float32(a.tm_stddev) * np.float32(std.value)
            rows.append(dict(word=w, direction=direction, gc=s["gc"][i], mean=np.float32(mean), std=np.float32(std.value),
                             tm=tm[i], tm_ok=bool(tm_ok), any=s["any"][i], end=s["end"][i], hp=s["hp"][i],
                             runs=bool(host.odm_is_run(w.encode()))))
        return rows

    def filt(rows):      # main.rs:492-516
        if a.keep_all == "true":
            return rows
        f32 = np.float32
        out = []
        for s in rows:
            ok = ((a.check_self_dimers != "true" or s["any"] < f32(a.max_self_dimer_any_tm)) and
                  (a.check_self_dimers != "true" or s["end"] < f32(a.max_self_dimer_end_tm)) and
                  (a.check_hairpin != "true" or s["hp"] < f32(a.max_hairpin_tm)) and
                  (a.disable_min_max_tm == "true" or (s["tm"] > f32(a.min_tm) and s["tm"] < f32(a.max_tm))) and
                  (a.disable_tm_stddev == "true" or s["tm_ok"]) and not s["runs"])
            if ok:
                out.append(s)
        return out

    prim_f, prim_r = filt(kmer_stats(cand_f, 0)), filt(kmer_stats(cand_r, 1))
    primers = [s["word"] for s in prim_f] + [s["word"] for s in prim_r]

    # ---- stage C: row blocks of the pool, edges to rank 0 -----------------------------------------
    nodes = list(dict.fromkeys(primers))       # the reference's graph is keyed by the primer string
    edges_txt = []
    if a.check_cross_dimers == "true" and nodes:
        n = len(nodes)
        # the candidates arrive ordered by frequency, direction by direction: rows are dealt out in groups of 256, round
        # robin (distributed.dealt_rows = msspe_group_rows), not cut into contiguous blocks
        rows = dealt_rows(n, world, rank)
        chem = capi.Chem(_two(a.mv_conc), _two(a.dv_conc), _two(a.dntp_conc), _two(a.dna_conc), _two(a.annealing_temp), 30)
        d_pool = torch.from_numpy(capi.pack_oligos(nodes).view(np.int64)).to(dev)
        cap = max(4096, len(rows) * n // 16)
        mine = []
        while len(rows):
            mine, count = screen_dealt_rows_edges(eng, d_pool, rows, k, chem, float(np.float32(a.delta_g_threshold)), cap)
            if count > cap:
                cap = count
                continue
            break
        gathered = [mine]
        if world > 1:
            gathered = [None] * world
            dist.all_gather_object(gathered, mine)
        if rank == 0:
            rc_cache = {}

            def revcomp(s):
                if s not in rc_cache:
                    rc_cache[s] = s[::-1].translate(str.maketrans("ACGTU", "TGCAA"))
                return rc_cache[s]
            # by (a, b), as the reference's nested loops emit the pairs (delta_g.rs:64-78), whoever screened the row
            for ia, ib in sorted(e for part in gathered for e in part):
                x, y = nodes[ia], nodes[ib]
                if a.check_self_dimers != "true" and (x == y or revcomp(y) == x):   # delta_g.rs:66-69
                    continue
                edges_txt.append(f"{x},{y}")

    rc = 0
    if rank == 0:
        deleted = set()
        if edges_txt:
            code, out = _call(host.odm_vertex_cover, "\n".join(primers).encode(), "\n".join(edges_txt).encode())
            deleted = set(out.split())
        keep = a.keep_all == "true"
        rows = [s for s in prim_f + prim_r if keep or s["word"] not in deleted]
        text = "\n".join(f'{s["word"]},{s["direction"]},{float(s["gc"]):.9g},{float(s["mean"]):.9g},'
                         f'{float(s["std"]):.9g},{float(s["tm"]):.9g}' for s in rows)
        code, csv = _call(host.odm_primers_csv, text.encode())
        Path(a.output).write_text(csv)
        print(f"pipeline_ranks: {world} rank(s), {len(cand_f)}+{len(cand_r)} candidates, {len(primers)} after the filters, "
              f"{len(edges_txt)} conflict edges, {len(rows)} primers written to {a.output}")
    eng.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return rc


if __name__ == "__main__":
    sys.exit(main())
