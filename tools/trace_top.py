#!/usr/bin/env python3
"""Top kernels of the LAST `--window` seconds of a rocprofv3 --kernel-trace CSV (warm-up and MIOpen's algorithm search excluded).
    python tools/trace_top.py <dir or kernel_trace.csv> [--window 0.7] [--steps N]"""
import argparse
import collections
import csv
import glob
import os
import re

ap = argparse.ArgumentParser()
ap.add_argument("path")
ap.add_argument("--window", type=float, default=0.7, help="seconds before the last kernel's end")
ap.add_argument("--steps", type=float, default=None, help="steps inside the window (prints per-step figures)")
ap.add_argument("--top", type=int, default=40)
a = ap.parse_args()
f = a.path if a.path.endswith(".csv") else sorted(glob.glob(os.path.join(a.path, "**", "*kernel_trace.csv"), recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
t1 = max(int(r["End_Timestamp"]) for r in rows)
cut = t1 - int(a.window * 1e9)
agg = collections.defaultdict(lambda: [0, 0])
for r in rows:
    s = int(r["Start_Timestamp"])
    if s < cut:
        continue
    n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
    n = re.sub(r"^void ", "", n)[:72]
    agg[n][0] += int(r["End_Timestamp"]) - s
    agg[n][1] += 1
tot = sum(v[0] for v in agg.values())
per = f", {tot / 1e6 / a.steps:.1f} ms per step" if a.steps else ""
print(f"{f}: last {a.window} s: kernels busy {tot / 1e6:.1f} ms ({100 * tot / (a.window * 1e9):.0f} % of the window{per}), {sum(v[1] for v in agg.values())} launches")
for n, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:a.top]:
    print(f"{v[0] / 1e6:8.2f} ms {100 * v[0] / tot:5.1f} % {v[1]:6d} x {v[0] / v[1] / 1e3:9.1f} us  {n}")
