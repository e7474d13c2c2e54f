#!/usr/bin/env python3
"""torch.profiler summary of one train_full_v2 / train_bev_v2 step (GPU)."""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lav_amd.train import LAV, TrainConfig, synthetic_bev_batch, synthetic_lidar_batch  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "lidar"
if len(sys.argv) > 2 and sys.argv[2] == "bench":
    torch.backends.cudnn.benchmark = True
dev = torch.device("cuda")
lav = LAV(TrainConfig(log_every=int(os.environ.get("LOG_EVERY", "100"))), dev, what=what)
B = int(os.environ.get("BATCH", "4" if what == "lidar" else "8"))
batch = synthetic_lidar_batch(B, device=dev) if what == "lidar" else synthetic_bev_batch(B, device=dev)
step = (lambda: lav.train_lidar(*batch)) if what == "lidar" else (lambda: lav.train_bev(*batch, other_weight=0.5))
for _ in range(3):
    step()
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(5):
    step()
torch.cuda.synchronize()
print(f"step {(time.perf_counter() - t0) / 5 * 1e3:.1f} ms")
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by=os.environ.get("SORT", "cuda_time_total"), row_limit=30, max_name_column_width=60))
