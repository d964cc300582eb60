"""Pure-Python restatement of od-msspe's outer pipeline (main.rs:596-861) on top of the C oracle.
TEST INFRASTRUCTURE ONLY: small inputs, used by tests/ to check the host layer's CSV and report.

Follows /root/reference/od-msspe/src/main.rs:108-122 (to_records), :408-516 (stats + filter),
:739-825 (cross-dimer graph + greedy vertex cover), :518-594 (coverage report), :834-858 (CSV)
and delta_g.rs:61-81 (which ordered pairs are sent to ntthal).
"""
from __future__ import annotations

import numpy as np

import pyoracle as o


def to_records(fasta: str):
    recs, name, seq = [], None, []
    for line in fasta.splitlines():
        line = line.rstrip("\r")
        if line.startswith(">"):
            if name is not None:
                recs.append((name, "".join(seq)))
            name, seq = line[1:].split(" ")[0], []
        elif name is not None:
            seq.append(line.upper().replace("U", "T"))
    if name is not None:
        recs.append((name, "".join(seq)))
    return recs


def f2(x) -> str:
    return "%.2f" % float(np.float32(x))


def vertex_cover(primers, edges):
    """main.rs:754-798.  edges: set of directed (a, b) conflicts."""
    conflicts = {}
    for p in primers:
        for (a, b) in edges:
            if a == p or b == p:
                conflicts.setdefault(a, set()).add(b)
                conflicts.setdefault(b, set()).add(a)
    deleted = set()
    while True:
        worst = None
        for p, nb in conflicts.items():
            if p in deleted:
                continue
            active = sum(1 for x in nb if x not in deleted)
            if active > 0 and (worst is None or (active, p) > worst):
                worst = (active, p)
        if worst is None:
            return deleted
        deleted.add(worst[1])


def coverage_report(fwd, rev, recs, seg, stride, win, k):
    sel_f, sel_r = set(fwd), set(rev)
    total = covered = 0
    seq_stats, part_stats = {}, {}
    for name, s in recs:
        for j, part in enumerate(o.partitions(s, seg, stride)):
            hit = any(w in sel_f for w in o.find_kmers(part[:win], k)) or \
                  any(o.reverse_complement(w) in sel_r for w in o.find_kmers(part[len(part) - win:], k))
            se = seq_stats.setdefault(name, [0, 0])
            pe = part_stats.setdefault(j & 0xFFFF, [0, 0])
            se[1] += 1
            pe[1] += 1
            total += 1
            if hit:
                se[0] += 1
                pe[0] += 1
                covered += 1
    covs = [np.float32(c) / np.float32(t) * np.float32(100.0) for c, t in seq_stats.values()]
    out = "\nCoverage report:\n"
    out += "  Segments:  %d/%d covered (%.1f%%)\n" % (
        covered, total, float(np.float32(100.0) * np.float32(covered) / np.float32(total)))
    out += "  Sequences: %d/%d at ≥80%% coverage (min %.1f%%, max %.1f%%)\n" % (
        sum(1 for c in covs if c >= 80.0), len(seq_stats), float(min(covs)), float(max(covs)))
    unc = sorted(p for p, (c, _) in part_stats.items() if c == 0)
    out += "  All partitions have primer coverage\n" if not unc else \
        "  Uncovered partitions: [%s]\n" % ", ".join(str(p) for p in unc)
    return out


def run(fasta: str, *, kmer_size=13, window_size=500, overlap_size=250, search_windows_size=50,
        max_iterations=1000, max_mismatch_segments=None, keep_all=False, check_cross_dimers=True,
        check_self_dimers=True, check_hairpin=True, tm_stddev=2.0, disable_tm_stddev=False,
        disable_min_max_tm=False, min_tm=30.0, max_tm=60.0, max_any=47.0, max_end=47.0, max_hairpin=24.0,
        mv=50.0, dv=3.0, dntp=0.0, dna=250.0, temp=25.0, dg=-9000.0, sample_stddev=True, candidates=None):
    """candidates: optional {direction: [(word, frequency), ...]} already produced by Segments.candidates with
    the same options (tools/make_config2_fixture.py runs stage A once and keeps the lists)."""
    recs = to_records(fasta)
    tables = o.Tables()
    mm = max_mismatch_segments if max_mismatch_segments is not None else min(10, max(1, -(-len(recs) // 50)))
    segs = None if candidates is not None else \
        o.Segments([s for _, s in recs], window_size, overlap_size, search_windows_size, kmer_size)
    f32 = np.float32
    result = {}
    for d in (0, 1):
        cand = candidates[d] if candidates is not None else segs.candidates(d, max_iterations, mm)
        words = [w for w, _ in cand]
        stats = []
        if words:
            info = o.check_primers(tables, words)
            mean, std = o.tm_stat(info["tm_f32"], sample_stddev)
            for i, w in enumerate(words):
                tm = f32(info["tm_f32"][i])
                st = dict(word=w, gc=f32(info["gc_f32"][i]), mean=f32(mean), std=f32(std), tm=tm,
                          tm_ok=bool(abs(tm - f32(mean)) <= f32(tm_stddev) * f32(std)),
                          any=f32(info["self_any_f32"][i]), end=f32(info["self_end_f32"][i]),
                          hp=f32(info["hairpin_f32"][i]), runs=o.is_run(w))
                stats.append(st)
        if not keep_all:
            stats = [s for s in stats if
                     (not check_self_dimers or s["any"] < f32(max_any)) and
                     (not check_self_dimers or s["end"] < f32(max_end)) and
                     (not check_hairpin or s["hp"] < f32(max_hairpin)) and
                     (disable_min_max_tm or (s["tm"] > f32(min_tm) and s["tm"] < f32(max_tm))) and
                     (disable_tm_stddev or s["tm_ok"]) and not s["runs"]]
        result[d] = stats
    primers = [s["word"] for s in result[0]] + [s["word"] for s in result[1]]
    edges = set()
    uniq = list(dict.fromkeys(primers))
    if check_cross_dimers and uniq:
        args = o.ntthal_args(float("%.2f" % mv), float("%.2f" % dv), float("%.2f" % dntp),
                             float("%.2f" % dna), float("%.2f" % temp))
        _, _, cf, _ = o.pool_pairs(tables, uniq, args, dg, want_dg=False)
        for a in range(len(uniq)):
            for b in range(len(uniq)):
                if cf[a, b]:
                    if not check_self_dimers and (uniq[a] == uniq[b] or
                                                  o.reverse_complement(uniq[b]) == uniq[a]):
                        continue
                    edges.add((uniq[a], uniq[b]))
    deleted = vertex_cover(primers, edges)
    good = {d: [s for s in result[d] if keep_all or s["word"] not in deleted] for d in (0, 1)}
    csv = "direction,name,primers,gc,avg,std,tm\n"
    for d, tag in ((0, "F"), (1, "R")):
        for idx, s in enumerate(good[d]):
            csv += f'{tag},Primer_{idx}_{tag},{s["word"]},{f2(s["gc"] / f32(100.0))},{f2(s["mean"])},' \
                   f'{f2(s["std"])},{f2(s["tm"])}\n'
    report = coverage_report([s["word"] for s in good[0]], [s["word"] for s in good[1]], recs,
                             window_size, overlap_size, search_windows_size, kmer_size)
    return csv, report, dict(candidates={d: len(result[d]) for d in (0, 1)}, deleted=deleted, edges=edges)
