#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV of `bench.py --shard-of 8 ...`: durations of the pass's kernels and the period
between consecutive filter launches over the last passes. usage: shard_timeline.py <kernel_trace.csv> [label] [passes to skip at the end]"""
import csv
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    tag = "F" if "k_fr_filter_q8" in n else "R" if "k_fr_refine" in n else "Z" if "k_finalize" in n else None
    if tag:
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), tag))
rows.sort()
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 0  # passes to leave out at the end (bench.py's serial probe)
if skip:
    rows = rows[:-3 * skip]
rows = rows[-3 * 150:]
label = sys.argv[2] if len(sys.argv) > 2 else ""
for tag in "FRZ":
    d = [(e - s) / 1e3 for s, e, t in rows if t == tag]
    if d:
        print("%s durations %s mean %.2f min %.2f us over %d launches" % (label, tag, sum(d) / len(d), min(d), len(d)))
f = [s for s, e, t in rows if t == "F"]
per = [(b - a) / 1e3 for a, b in zip(f, f[1:])]
if per:
    per.sort()
    print("%s F period mean %.2f median %.2f min %.2f us" % (label, sum(per) / len(per), per[len(per) // 2], per[0]))
t0 = rows[-12][0]
for s, e, t in rows[-12:]:
    print("    %s start %8.2f end %8.2f" % (t, (s - t0) / 1e3, (e - t0) / 1e3))
