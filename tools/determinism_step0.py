#!/usr/bin/env python3
"""Step 0 of train_lidar N times from the same state, deterministic switches on: which parameter gradients ever differ between runs,
and how often (round 6: across processes the 500-step curve is a random sample - the first difference appears at step 0, 1 or 2).
    [LAV_TRAIN_...=torch] python tools/determinism_step0.py [runs=16]"""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lav_amd.train import LAV, TrainConfig  # noqa: E402
from lav_amd.train.run import set_deterministic  # noqa: E402
from lav_amd.train.synthetic import synthetic_lidar_batch  # noqa: E402

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 16
set_deterministic(True)
dev = torch.device("cuda")
batch = synthetic_lidar_batch(2, seed=40, max_points=20000, num_objs=3)
ref, differ, nbad = None, collections.Counter(), 0
for r in range(runs):
    torch.manual_seed(0)
    lav = LAV(TrainConfig(log_inference=False), dev, what="lidar")
    torch.manual_seed(1000)
    info = lav.train_lidar(*batch)
    g = {n: p.grad.detach().clone() for n, p in lav.student.named_parameters() if p.grad is not None}
    if ref is None:
        ref = g
        continue
    bad = [n for n in g if not torch.equal(g[n], ref[n])]
    nbad += bool(bad)
    for n in bad:
        differ[".".join(n.split(".")[:4])] += 1
    if bad:
        worst = max(bad, key=lambda n: ((g[n] - ref[n]).abs().max() / ref[n].abs().max().clamp_min(1e-30)).item())
        print(f"run {r}: {len(bad)} of {len(g)} gradients differ from run 0; worst {worst} rel {((g[worst] - ref[worst]).abs().max() / ref[worst].abs().max()).item():.2e}", flush=True)
print(f"{nbad} of {runs - 1} runs differ from run 0")
for k, v in sorted(differ.items()):
    print(f"  {k:60s} in {v} runs")
