"""GPU idle gaps inside one frame of a rocprofv3 --kernel-trace CSV (all streams): the union of kernel intervals between two
consecutive k_merge_ticks launches of the replayed lidar graph, every idle gap > 2 us with the kernels around it.
    python tools/trace_gaps.py <trace dir> [frame index from the start = 30]"""
import collections
import csv
import glob
import os
import re
import sys

d = sys.argv[1]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 30
rows = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    rows += list(csv.DictReader(open(f)))
for r in rows:
    r["_s"], r["_e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["_s"])
sh = lambda n: re.sub(r"\(.*", "", n.replace("(anonymous namespace)::", "").replace("void ", ""))[:60]
starts = [i for i, r in enumerate(rows) if "k_merge_ticks" in r["Kernel_Name"]]
# (bench.py ends with stand-alone replays of every graph: frames are counted from the START: 16-20 warm-up frames, then the timed ones)
lo, hi = starts[back], starts[back + 1]
win = rows[lo:hi]
t0, t1 = win[0]["_s"], rows[hi]["_s"]
print(f"# frame window {(t1 - t0) / 1e3:.1f} us, {len(win)} kernels on {len(set(r['Stream_Id'] for r in win))} streams")
busy_end, idle, gaps = t0, 0, []
prev = None
for r in win:
    if r["_s"] > busy_end:
        g = r["_s"] - busy_end
        idle += g
        if g > 2000:
            gaps.append((g / 1e3, (busy_end - t0) / 1e3, sh(prev["Kernel_Name"]) if prev else "-", sh(r["Kernel_Name"]), r["Stream_Id"]))
    if r["_e"] > busy_end:
        busy_end, prev = r["_e"], r
print(f"# GPU idle (no kernel of any stream running): {idle / 1e3:.1f} us of {(t1 - t0) / 1e3:.1f}")
for g, at, a, b, st in gaps:
    print(f"  gap {g:7.1f} us at {at:8.1f}: after {a} -> before {b} (stream {st})")
per = collections.defaultdict(lambda: [0, 0.0])
for r in win:
    a = per[r["Stream_Id"]]; a[0] += 1; a[1] += (r["_e"] - r["_s"]) / 1e3
for st, (n, t) in sorted(per.items()):
    print(f"# stream {st}: {n} kernels, {t:.1f} us busy")
tiny = collections.Counter()
for r in win:
    if r["_e"] - r["_s"] < 8000:
        tiny[sh(r["Kernel_Name"])] += 1
print("# kernels shorter than 8 us in the window:", dict(tiny.most_common(20)))
