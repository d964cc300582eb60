#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output for one kernel (mean per dispatch)."""
import collections
import csv
import glob
import sys

pattern, kernel = sys.argv[1], sys.argv[2]
tot = collections.defaultdict(list)
dur = []
for f in glob.glob(pattern, recursive=True):
    for r in csv.DictReader(open(f)):
        if kernel not in r["Kernel_Name"]:
            continue
        tot[r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
for c in sorted(tot):
    v = tot[c]
    print(f"{c:28s} n={len(v):3d} mean={sum(v)/len(v):.6g}")
if dur:
    print(f"dispatch_ms mean={sum(dur)/len(dur):.3f}")
