#!/usr/bin/env python3
"""Turns gpurun_out/profiles_raw/ (tools/collect_profiles.sh) into the tracked files under profiles/."""
import collections
import csv
import glob
import json
import shutil
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
RAW = ROOT / "gpurun_out" / "profiles_raw"
OUT = ROOT / "profiles"
ROUND = sys.argv[1] if len(sys.argv) > 1 else "r01"
kProbeChecks = 65536.0 * 65536.0       # the counter passes run the bench itself: one step of the 65,536-primer pool
kLaunchesPerStep = 9.0                 # ... which the engine cuts into nine equal first-stage launches (capi.cpp kChunkPairs)
kLaunchChecks = kProbeChecks / kLaunchesPerStep   # the launch every per-launch figure below belongs to
KERNEL = "k_pairs_row"
KERNEL_MATCH = "k_pairs_row<"   # the row-specialised first stage (thal_pairs_row.hip)
# the commit whose kernels were measured: the last one that touched the kernel sources (pass it as argv[2] when
# the working tree had uncommitted kernel changes at collection time)
COMMIT = sys.argv[2] if len(sys.argv) > 2 else subprocess.run(
    ["git", "log", "-1", "--format=%h", "--", "open-msspe-design_amd/csrc"], cwd=ROOT, capture_output=True,
    text=True).stdout.strip()


def pmc(sub):
    tot = collections.defaultdict(list)
    dur = []
    for f in glob.glob(str(RAW / sub / "**" / "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if KERNEL_MATCH not in r["Kernel_Name"]:
                continue
            tot[r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    n = len(dur) // max(len(tot), 1)
    # mean per dispatch, as dispatched: the step's launches are equal (a multiple of kLaunchesPerStep of them)
    assert n % int(kLaunchesPerStep) == 0, f"{sub}: {n} dispatches, expected a multiple of {kLaunchesPerStep}"
    return ({k: sum(v) / len(v) for k, v in tot.items()}, (sum(dur) / len(dur) if dur else 0.0), n)


shutil.copy(RAW / "bench_stats" / "bench_kernel_stats.csv", OUT / f"{ROUND}_bench_kernel_stats.csv")
bd = RAW / "bench_default_stats" / "benchdef_kernel_stats.csv"
if bd.exists():
    shutil.copy(bd, OUT / f"{ROUND}_bench_default_kernel_stats.csv")
    dl = [l for l in open(RAW / "bench_default_stdout.log") if l.startswith('{"metric"')]
    if dl:
        (OUT / f"{ROUND}_bench_default_line.json").write_text(dl[-1])
shutil.copy(RAW / "stage_a_stats" / "stagea_kernel_stats.csv", OUT / f"{ROUND}_stage_a_kernel_stats.csv")
(OUT / f"{ROUND}_stage_a_kernel_stats.commit").write_text(COMMIT + "\n")   # bench.py names it beside the imported figure
sp = RAW / "small_pool_stats" / "smallpool_kernel_stats.csv"
if sp.exists():
    shutil.copy(sp, OUT / f"{ROUND}_small_pool_kernel_stats.csv")
for sub, name in (("stage_b_stats", "stageb"), ("stage_b2k_stats", "stageb2k")):
    src = RAW / sub / f"{name}_kernel_stats.csv"
    if src.exists():
        shutil.copy(src, OUT / f"{ROUND}_{'stage_b_1m' if name == 'stageb' else 'stage_b_2000'}_kernel_stats.csv")
line = [l for l in open(RAW / "bench_stdout.log") if l.startswith('{"metric"')][-1]
(OUT / f"{ROUND}_bench_line.json").write_text(line)

c1, ms1, n1 = pmc("pmc1")
c2, ms2, n2 = pmc("pmc2")
cf, msf, nf = pmc("pmc_fetch")
cw, msw, nw = pmc("pmc_write")
checks = kLaunchChecks
waves = checks / 64.0
clock_ghz = c2["GRBM_GUI_ACTIVE"] / 8.0 / (ms2 * 1e-3) / 1e9
simd_quads = 1024.0 * c2["GRBM_GUI_ACTIVE"] / 8.0 / 4.0
wave_slots = 1024.0 * 3.0   # 768-thread blocks: three waves per SIMD
with open(OUT / f"{ROUND}_pmc_{KERNEL}.txt", "w") as f:
    f.write(f"# rocprofv3 --pmc (separate passes, tools/collect_profiles.sh) -- python3 bench.py --steps 1 --warmup 0 (the bench's own 65,536-primer pool); kernel {KERNEL},\n"
            f"# mean per dispatch ({checks:.6g} checks = a ninth of the step = {int(waves)} wave-batches of 64 pairs). SQ_* cycle counters are in quad-cycles;\n"
            f"# GRBM_GUI_ACTIVE sums the 8 XCDs (/8 = {c2['GRBM_GUI_ACTIVE']/8e6:.1f} M cycles in {ms2:.2f} ms = {clock_ghz:.2f} GHz)\n")
    for name, (c, ms, n) in (("pass 1", (c1, ms1, n1)), ("pass 2", (c2, ms2, n2)), ("FETCH_SIZE [KB]", (cf, msf, nf)),
                             ("WRITE_SIZE [KB]", (cw, msw, nw))):
        f.write(f"## {name}: {n} dispatches, {ms:.3f} ms each\n")
        for k in sorted(c):
            f.write(f"{k:28s} {c[k]:.6g}\n")
    f.write("## derived\n")
    f.write(f"VALU instructions per check (per lane)      {c1['SQ_INSTS_VALU'] / waves:.0f}\n")
    f.write(f"SALU / LDS / branch per wave-batch          {c1['SQ_INSTS_SALU'] / waves:.0f} / {c1['SQ_INSTS_LDS'] / waves:.0f} / {c1['SQ_INSTS_BRANCH'] / waves:.0f}\n")
    per_simd_cycles = c2['GRBM_GUI_ACTIVE'] / 8.0
    f.write(f"cycles per VALU instruction per SIMD        {per_simd_cycles / (c1['SQ_INSTS_VALU'] / 1024.0):.3f}"
            f"   (measured issue limits, profiles/{ROUND}_valu_peak.txt: 2.4 for add/sub/and/or/lshr/mov, 4.3 for the rest, "
            f"v_cmp 5.5 at three waves per SIMD)\n")
    f.write(f"wave cycles: active / issue-stalled / waiting   {c2['SQ_ACTIVE_INST_ANY'] / c1['SQ_WAVE_CYCLES']:.3f} / "
            f"{c1['SQ_WAIT_INST_ANY'] / c1['SQ_WAVE_CYCLES']:.3f} / {c1['SQ_WAIT_ANY'] / c1['SQ_WAVE_CYCLES']:.3f}\n")
    f.write(f"LDS bank-conflict share of LDS active       {c2['SQ_LDS_BANK_CONFLICT'] / c2['SQ_LDS_IDX_ACTIVE']:.3f}\n")
    f.write(f"SQ_ACTIVE_INST_VALU x 4 / SIMD cycles         {c2['SQ_ACTIVE_INST_VALU'] / simd_quads:.3f}   (SQ_ACTIVE_INST_VALU = SQ_INSTS_VALU x "
            f"{c2['SQ_ACTIVE_INST_VALU'] / c1['SQ_INSTS_VALU']:.3f}: on gfx950 the counter repeats the instruction count -- a value above 1 here "
            f"means more 'busy quad-cycles' than the launch has -- so it is NOT a busy-time measure; the figure to read is cycles per instruction above)\n")
hbm = 2.0 * cf["FETCH_SIZE"] * 1024.0 + cw["WRITE_SIZE"] * 1024.0
(OUT / "traffic_latest.json").write_text(json.dumps({
    "kernel": KERNEL, "round": int(ROUND[1:]), "commit": COMMIT,
    "command": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python3 bench.py --steps 1 --warmup 0 --no-stage-a --no-stage-b --no-cpu-baseline --no-small-pool",
    "launches_sampled": nf,
    "FETCH_SIZE_KB_per_launch": cf["FETCH_SIZE"], "WRITE_SIZE_KB_per_launch": cw["WRITE_SIZE"],
    "correction": "gfx950: FETCH_SIZE under-reports wide coalesced reads by 2x (MI355X_MICROARCH.md HBM section) -> doubled; WRITE_SIZE taken as is",
    "hbm_bytes_per_launch": hbm, "checks_per_launch": checks,
    "algorithmic_bytes_per_launch": 8.0 * (65536.0 / kLaunchesPerStep + 65536.0) + checks / 8.0 + 4.0 * 65536.0 / kLaunchesPerStep,
    "note": "per first-stage launch as dispatched (a ninth of the 65,536^2 step); traffic = the 64-byte atomics that set "
            "conflict bits (0.5 % of pairs), the list of pairs handed to the later stages (about 2 %, 8 B each), the "
            "per-pair register spills (two registers, outside the cell loop) and the per-row table builds; bench.py scales "
            "the bytes per check to its own launch"}, indent=1))
(OUT / "pmc_latest.json").write_text(json.dumps({
    "kernel": KERNEL, "source": f"profiles/{ROUND}_pmc_{KERNEL}.txt",
    "valu_instructions_per_check": c1["SQ_INSTS_VALU"] / waves,
    "salu_instructions_per_wave_batch": c1["SQ_INSTS_SALU"] / waves,
    "lds_instructions_per_wave_batch": c1["SQ_INSTS_LDS"] / waves,
    "cycles_per_valu_instruction_per_simd": (c2["GRBM_GUI_ACTIVE"] / 8.0) / (c1["SQ_INSTS_VALU"] / 1024.0),
    "measured_issue_limit_cycles": {"add/sub/and/or/lshr/mov": 2.4, "other VALU": 4.3, "v_cmp at 3 waves/SIMD": 5.5,
                                    "source": f"profiles/{ROUND}_valu_peak.txt"},
    "wave_cycle_shares": {"active": c2["SQ_ACTIVE_INST_ANY"] / c1["SQ_WAVE_CYCLES"],
                          "issue_stalled": c1["SQ_WAIT_INST_ANY"] / c1["SQ_WAVE_CYCLES"],
                          "waiting": c1["SQ_WAIT_ANY"] / c1["SQ_WAVE_CYCLES"]},
    "lds_bank_conflict_share": c2["SQ_LDS_BANK_CONFLICT"] / c2["SQ_LDS_IDX_ACTIVE"],
    "achieved_int32_tops": c1["SQ_INSTS_VALU"] * 64 / (ms1 * 1e-3) / 1e12,
    "clock_ghz": clock_ghz, "commit": COMMIT, "round": int(ROUND[1:])}, indent=1))
# VALU issue-rate microbenchmark -> readable table
vp = RAW / "valu_peak.jsonl"
if vp.exists():
    rows = collections.OrderedDict()
    head = ""
    for l in open(vp):
        try:
            d = json.loads(l)
        except Exception:
            continue
        if "op" not in d:
            head = json.dumps(d)
            continue
        rows.setdefault(d["op"], {})[d["waves_per_simd"]] = d["cycles_per_valu_per_simd"]
    shutil.copy(vp, OUT / f"{ROUND}_valu_peak.jsonl")
    with open(OUT / f"{ROUND}_valu_peak.txt", "w") as f:
        f.write("# tools/valu_peak (tools/valu_peak.hip): cycles per wave64 VALU instruction per SIMD, independent chains,\n"
                "# every CU busy, 1..4 waves per SIMD; cycles = launch wall time x in-kernel clock (s_memtime / s_memrealtime)\n"
                f"# {head}\n")
        f.write(f"{'op':30s} {'w=1':>6s} {'w=2':>6s} {'w=3':>6s} {'w=4':>6s}\n")
        for k, v in rows.items():
            f.write(f"{k:30s} " + " ".join(f"{v.get(w, float('nan')):6.2f}" for w in (1, 2, 3, 4)) + "\n")
print(open(OUT / f"{ROUND}_pmc_{KERNEL}.txt").read())
