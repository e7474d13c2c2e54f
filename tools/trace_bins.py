"""GPU busy fraction over time from a rocprofv3 --kernel-trace CSV: the last `--window` seconds cut into bins of `--bin` ms, each with its
busy share (union of kernel intervals), launch count and the kernels that took most of it - where a training step leaves the GPU idle.
    python tools/trace_bins.py <trace dir> [--window 0.25] [--bin 2]"""
import argparse
import collections
import csv
import glob
import os
import re

ap = argparse.ArgumentParser()
ap.add_argument("path")
ap.add_argument("--window", type=float, default=0.25)
ap.add_argument("--bin", type=float, default=2.0)
a = ap.parse_args()
rows = []
for f in glob.glob(os.path.join(a.path, "**", "*kernel_trace.csv"), recursive=True):
    rows += list(csv.DictReader(open(f)))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
t_end = max(e for _, e, _ in ev)
t0 = t_end - int(a.window * 1e9)
ev = [x for x in ev if x[1] > t0]
sh = lambda n: re.sub(r"\(.*", "", n.replace("(anonymous namespace)::", "").replace("void ", "").replace("at::native::", ""))[:44]
nb = int(a.window * 1e3 / a.bin) + 1
busy = [0.0] * nb
cnt = [0] * nb
names = [collections.Counter() for _ in range(nb)]
cur = t0
for s, e, n in ev:
    s2 = max(s, cur, t0)          # union of intervals: time already covered by an earlier kernel is not counted twice
    if e > s2:
        b = int((s2 - t0) / 1e6 / a.bin)
        while s2 < e and b < nb:
            edge = t0 + int((b + 1) * a.bin * 1e6)
            seg = min(e, edge) - s2
            busy[b] += seg
            names[b][sh(n)] += seg
            s2 += seg
            b += 1
        cur = max(cur, e)
    b0 = int((max(s, t0) - t0) / 1e6 / a.bin)
    if b0 < nb:
        cnt[b0] += 1
for b in range(nb):
    top = ", ".join(f"{k} {v / 1e3:.0f}us" for k, v in names[b].most_common(3))
    print(f"{b * a.bin:7.1f} ms  busy {busy[b] / (a.bin * 1e6) * 100:5.1f} %  {cnt[b]:4d} launches  {top}")
