#!/usr/bin/env python3
"""Kernel sequence of ONE frame out of a rocprofv3 --kernel-trace CSV of `bench.py --steps K` (no micro-benchmarks after the
frames would be cleaner: pass --frame-from-end to pick a frame before them): launches in start order with stream, duration and
gap to the previous kernel on the same stream.
    python tools/frame_sequence.py <dir> [--anchor k_bin] [--index -3]"""
import argparse
import csv
import glob
import os
import re

ap = argparse.ArgumentParser()
ap.add_argument("path")
ap.add_argument("--anchor", default="merge_ticks", help="kernel that starts a frame")
ap.add_argument("--index", type=int, default=-40, help="which occurrence of the anchor (negative: from the end)")
a = ap.parse_args()
f = sorted(glob.glob(os.path.join(a.path, "**", "*kernel_trace.csv"), recursive=True))[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
short = lambda n: re.sub(r"\(anonymous namespace\)::|^void ", "", n)[:48]
starts = [i for i, r in enumerate(rows) if a.anchor in r["Kernel_Name"]]
i0 = starts[a.index]
i1 = starts[a.index + 1] if a.index + 1 < 0 or a.index + 1 < len(starts) else len(rows)
t0 = int(rows[i0]["Start_Timestamp"])
last_end = {}
print(f"{f}: frame of {i1 - i0} launches, {(int(rows[i1 - 1]['End_Timestamp']) - t0) / 1e3:.0f} us")
for r in rows[i0:i1]:
    s, e, q = int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Stream_Id", r.get("Queue_Id", "?"))
    gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
    last_end[q] = e
    print(f"{(s - t0) / 1e3:8.1f} us  q{q:>3}  {(e - s) / 1e3:7.1f} us  gap {gap:6.1f}  {short(r['Kernel_Name'])}")
