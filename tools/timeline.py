#!/usr/bin/env python3
"""Per-queue busy segments of one frame from a rocprofv3 kernel trace (csv)."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
paints = [i for i, r in enumerate(rows) if 'k_paint' in r['Kernel_Name']]
a, b = paints[-4], paints[-3]
t0 = int(rows[a]['Start_Timestamp']); t1 = int(rows[b]['Start_Timestamp'])
fr = [r for r in rows if t0 <= int(r['Start_Timestamp']) < t1]
print('frame period us', (t1 - t0) / 1e3, 'kernels', len(fr))


def short(n):
    return n.replace('(anonymous namespace)::', '').replace('void ', '')[:26]


for q in sorted(set(r['Queue_Id'] for r in fr)):
    print('=== queue', q)
    prev_end = None; seg_start = None; cnt = 0; names = collections.Counter(); busy = 0
    for r in [r for r in fr if r['Queue_Id'] == q]:
        s = (int(r['Start_Timestamp']) - t0) / 1e3; e = (int(r['End_Timestamp']) - t0) / 1e3
        if prev_end is None or s - prev_end > 30:
            if prev_end is not None:
                print(f"  {seg_start:7.0f}-{prev_end:7.0f} us  n={cnt:3d} busy={busy:6.0f}  {names.most_common(3)}")
            seg_start = s; cnt = 0; names = collections.Counter(); busy = 0
        cnt += 1; names[short(r['Kernel_Name'])] += 1; prev_end = e; busy += e - s
    print(f"  {seg_start:7.0f}-{prev_end:7.0f} us  n={cnt:3d} busy={busy:6.0f}  {names.most_common(3)}")
