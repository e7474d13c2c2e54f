#!/usr/bin/env python3
"""One steady-state frame from a rocprofv3 results db: per-queue busy segments and the biggest gaps."""
import collections
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
marker = sys.argv[2] if len(sys.argv) > 2 else 'k_merge_ticks'
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]
ks = [t for t in tabs if 'kernel_symbol' in t][0]
rows = db.execute(f"select s.kernel_name, d.start, d.end, d.queue_id from {kd} d join {ks} s on d.kernel_id=s.id order by d.start").fetchall()
marks = [i for i, r in enumerate(rows) if marker in r[0]]
a, b = marks[-4], marks[-3]
t0, t1 = rows[a][1], rows[b][1]
fr = rows[a:b]
print('frame period us', (t1 - t0) / 1e3, 'kernels', len(fr), 'sum kernel us', sum(r[2] - r[1] for r in fr) / 1e3)


def short(n):
    n = n.replace('_ZN12_GLOBAL__N_1', '').replace('(anonymous namespace)::', '').replace('void ', '')
    return n[:30]


for q in sorted(set(r[3] for r in fr)):
    print('=== queue', q)
    prev_end = None; seg_start = None; cnt = 0; names = collections.Counter(); busy = 0
    for r in [r for r in fr if r[3] == q]:
        s = (r[1] - t0) / 1e3; e = (r[2] - t0) / 1e3
        if prev_end is None or s - prev_end > 30:
            if prev_end is not None:
                print(f"  {seg_start:7.0f}-{prev_end:7.0f} us  n={cnt:3d} busy={busy:6.0f}  {names.most_common(3)}")
            seg_start = s; cnt = 0; names = collections.Counter(); busy = 0
        cnt += 1; names[short(r[0])] += 1; prev_end = e; busy += e - s
    print(f"  {seg_start:7.0f}-{prev_end:7.0f} us  n={cnt:3d} busy={busy:6.0f}  {names.most_common(3)}")
if len(sys.argv) > 3:
    for r in fr:
        print(f"{(r[1]-t0)/1e3:8.1f} {(r[2]-r[1])/1e3:7.1f} q{r[3]} {short(r[0])}")
