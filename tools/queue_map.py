#!/usr/bin/env python3
"""Reads a rocprofv3 --kernel-trace CSV: which hardware queue every region-stage launch ran on, and the launches' durations.
   tools/queue_map.py <dir or kernel_trace.csv>"""
import csv, glob, os, sys
from collections import Counter, defaultdict
p = sys.argv[1]
f = p if p.endswith(".csv") else glob.glob(os.path.join(p, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "k_region" in r["Kernel_Name"]]
q = Counter(r["Queue_Id"] for r in rows)
dur = defaultdict(list)
for r in rows: dur[r["Queue_Id"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
print("%d region launches on %d queues" % (len(rows), len(q)))
for k in sorted(q, key=lambda x: int(x)): print("  queue %s: %d launches, mean %.1f ms" % (k, q[k], sum(dur[k]) / len(dur[k])))
