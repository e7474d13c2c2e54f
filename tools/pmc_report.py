#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc results db: mean counter value per kernel name (last 3 dispatches of each)."""
import collections
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
pmc = [t for t in tabs if 'pmc_event' in t][0]
info = [t for t in tabs if 'info_pmc' in t][0]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]
ks = [t for t in tabs if 'kernel_symbol' in t][0]
cols = [r[1] for r in db.execute(f"pragma table_info({pmc})")]
print("#", pmc, cols, file=sys.stderr)
q = (f"select s.kernel_name, d.dispatch_id, i.name, sum(e.value) from {pmc} e join {info} i on e.pmc_id=i.id "
     f"join {kd} d on e.event_id=d.event_id join {ks} s on d.kernel_id=s.id group by d.dispatch_id, i.name order by d.dispatch_id")
per = collections.defaultdict(lambda: collections.defaultdict(list))
for name, did, cname, val in db.execute(q):
    per[name][cname].append(val)
for name, cs in per.items():
    short = name.replace('(anonymous namespace)::', '')[:70]
    print(short, {c: round(sum(v[-3:]) / len(v[-3:]), 1) for c, v in cs.items()}, f"n={len(next(iter(cs.values())))}")
    if "k_rows" in name or "k_bin" in name:   # tools/pmc_pillar.py: the first half of the dispatches is the 32 769-point cloud
        print("    first cloud:", {c: round(sum(v[2:len(v) // 2]) / max(len(v[2:len(v) // 2]), 1), 1) for c, v in cs.items()})
