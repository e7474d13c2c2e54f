#!/usr/bin/env python3
"""Where a train_full step synchronises the host with the GPU (torch.cuda.set_sync_debug_mode("warn")) and how long the host needs to
enqueue a step (round 6: with the convolutions on fp16 pieces the step's kernels take 90 ms and the step 118 - the rest is the host).
    python tools/train_sync_probe.py [lidar|bev]"""
import collections
import os
import sys
import time
import traceback
import warnings

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from lav_amd.train import LAV, TrainConfig, synthetic_bev_batch, synthetic_lidar_batch  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "lidar"
dev = torch.device("cuda")
lav = LAV(TrainConfig(log_every=1), dev, what=what)
batch = synthetic_lidar_batch(32, seed=1, max_points=120000, num_objs=8) if what == "lidar" else synthetic_bev_batch(64, seed=1, num_objs=8)
step = lav.train_lidar if what == "lidar" else (lambda *b: lav.train_bev(*b, other_weight=0.5))
for _ in range(3):
    step(*batch)
torch.cuda.synchronize()
sites = collections.Counter()
orig = warnings.showwarning


def show(message, category, filename, lineno, file=None, line=None):
    if "synchroniz" in str(message):
        st = [f for f in traceback.extract_stack() if "/lav_amd/" in f.filename]
        sites[" <- ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in st[-3:])] += 1


warnings.showwarning = show
torch.cuda.set_sync_debug_mode("warn")
step(*batch)
torch.cuda.set_sync_debug_mode("default")
warnings.showwarning = orig
print("synchronisation sites of one step:")
for k, v in sites.most_common():
    print(f"  {v:3d} x {k}")
# host enqueue time vs step time
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(4):
    step(*batch)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"4 steps: host returned after {1e3 * (t1 - t0) / 4:.1f} ms per step, GPU done after {1e3 * (t2 - t0) / 4:.1f} ms per step")
