#!/usr/bin/env python3
"""cProfile of the HOST side of one train_full / train_bev step (round 6: the step's kernels take 90 ms, the step 117 - in the regions of
small layers the GPU waits for Python).    BATCH=32 python tools/train_cprofile.py [lidar|bev]"""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lav_amd.train import LAV, TrainConfig, synthetic_bev_batch, synthetic_lidar_batch  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "lidar"
dev = torch.device("cuda")
B = int(os.environ.get("BATCH", "32" if what == "lidar" else "64"))
lav = LAV(TrainConfig(log_every=1), dev, what=what)
batch = synthetic_lidar_batch(B, device=dev) if what == "lidar" else synthetic_bev_batch(B, device=dev)
step = (lambda: lav.train_lidar(*batch)) if what == "lidar" else (lambda: lav.train_bev(*batch, other_weight=0.5))
for _ in range(4):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(45)
st.sort_stats("cumulative").print_stats(60)
