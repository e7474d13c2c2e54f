#!/usr/bin/env python3
"""Workload for the PMC pass (rocprofv3 --pmc FETCH_SIZE WRITE_SIZE): calibration kernels of known byte counts
(26.2 MB fill, 26.2 MB copy) followed by the pillar pipeline at 32 769 and 196 608 points."""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import lav_amd  # noqa: E402
from lav_amd import synth  # noqa: E402

dev = torch.device("cuda:0")
cfg = dict(min_x=-10, max_x=70, min_y=-40, max_y=40, pixels_per_meter=4)
lm = lav_amd.LiDARModel(num_input=16, backbone="cnn", num_features=[64, 64], **cfg)
lm.load_state_dict(synth.seeded_state_dict(lm, prefix="lidar."))
ppn = lm.eval().to(dev).point_pillar_net
canvas = torch.empty((1, 64, 320, 320), device=dev)
src = torch.randn((1, 64, 320, 320), device=dev)
for _ in range(5):
    canvas.zero_()
    canvas.copy_(src)
for n in (10923, 65536):
    pts = torch.from_numpy(synth.stacked_lidar(n)).to(dev)
    for _ in range(5):
        ppn([pts], [len(pts)])
torch.cuda.synchronize()
