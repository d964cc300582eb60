#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output per kernel: mean counter value per dispatch, for the
dispatches whose kernel name contains the filter.  usage: pmc_summary.py '<glob>' <name filter>"""
import collections
import csv
import glob
import re
import sys

pattern, flt = sys.argv[1], sys.argv[2]
tot = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob(pattern, recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if flt not in name:
            continue
        short = re.sub(r"\(anonymous namespace\)::|msspe::|void ", "", name).split("(")[0]
        tot[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if "End_Timestamp" in r and r.get("End_Timestamp"):
            dur[short].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
for k in sorted(tot):
    print(f"== {k}")
    for c in sorted(tot[k]):
        v = tot[k][c]
        print(f"  {c:28s} n={len(v):3d} mean={sum(v)/len(v):.6g} max={max(v):.6g}")
    if dur[k]:
        print(f"  dispatch_ms mean={sum(dur[k])/len(dur[k]):.3f} max={max(dur[k]):.3f}")
