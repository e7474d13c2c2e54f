"""One frame of the agent out of a rocprofv3 --kernel-trace CSV: every kernel in start order with its queue, start (us from the
frame's first kernel), duration and the idle gap on its queue before it.

    rocprofv3 --kernel-trace -d gpurun_out/trace -o bench -- python bench.py --no-train --no-cpu-baseline --no-variants --steps 30
    python tools/trace_frame.py gpurun_out/trace [frame_from_end=3] > profiles/r04_frame_trace.txt
"""
import csv
import glob
import os
import re
import sys

d = sys.argv[1]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
rows = []
for f in files:
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append(r)
if not rows:
    sys.exit("no kernel trace rows under " + d)
k = rows[0].keys()
col = lambda *names: next(n for n in names if n in k)
c_name, c_start, c_end = col("Kernel_Name", "Name"), col("Start_Timestamp", "Start"), col("End_Timestamp", "End")
c_q = col("Queue_Id", "Queue")
for r in rows:
    r["_s"], r["_e"] = int(r[c_start]), int(r[c_end])
rows.sort(key=lambda r: r["_s"])


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"\(.*$", "", n)
    return n[:70]


c_st = "Stream_Id" if "Stream_Id" in k else c_q
import collections
main = collections.Counter(r[c_st] for r in rows if "k_merge_ticks" in r[c_name]).most_common(1)[0][0]   # (eager warm-ups run elsewhere)
side = [r for r in rows if r[c_st] != main]
rows = [r for r in rows if r[c_st] == main]        # the critical chain: the stream that runs the lidar / heads / others graphs
starts = [i for i, r in enumerate(rows) if "k_merge_ticks" in r[c_name]]
if len(starts) < back + 1:
    sys.exit(f"only {len(starts)} frames in the trace")
lo, hi = starts[-back - 1], starts[-back]
t0 = rows[lo]["_s"]
last_end = {}
print(f"# frame = kernels from one k_merge_ticks to the next: {hi - lo} kernels, {(rows[hi]['_s'] - t0) / 1e3:.1f} us")
print(f"# {'start':>8s} {'dur':>7s} {'gap':>7s}  queue  kernel")
per = {}
for r in rows[lo:hi]:
    q = r[c_q]
    gap = (r["_s"] - last_end[q]) / 1e3 if q in last_end else 0.0
    last_end[q] = r["_e"]
    nm = short(r[c_name])
    print(f"{(r['_s'] - t0) / 1e3:9.1f} {(r['_e'] - r['_s']) / 1e3:7.1f} {gap:7.1f}  {q:>5s}  {nm}")
    a = per.setdefault(nm, [0, 0.0])
    a[0] += 1; a[1] += (r["_e"] - r["_s"]) / 1e3
t1 = rows[hi]["_s"]
print("# side streams inside this frame window: per stream, kernels / busy us")
agg = {}
for r in side:
    if t0 <= r["_s"] < t1:
        a = agg.setdefault(r[c_st], [0, 0.0]); a[0] += 1; a[1] += (r["_e"] - r["_s"]) / 1e3
for st, (n, t) in agg.items():
    print(f"#   stream {st}: {n} kernels, {t:.1f} us")
print("# per kernel in this frame: calls, total us")
for nm, (n, t) in sorted(per.items(), key=lambda kv: -kv[1][1]):
    print(f"# {n:4d} {t:8.1f}  {nm}")
