#!/usr/bin/env python3
"""Time of lav_crop_rotate_backward at train_full's sizes (32 maps of 384 x 160 x 160, 96 crops of 96 x 96) and its
agreement with torch's grid_sample backward on a small case."""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from lav_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
M, C, H, W, n, crop = 32, 384, 160, 160, int(sys.argv[1]) if len(sys.argv) > 1 else 96, 96
g = torch.Generator().manual_seed(0)
feat = torch.randn((M, C, H, W), generator=g).to(dev).requires_grad_(True)
idx = (torch.arange(n) % M).int().to(dev)
locs = ((torch.rand((n, 2), generator=g) - 0.5) * 30).to(dev)
oris = ((torch.rand(n, generator=g) - 0.5) * 6.2).to(dev)
out = ops.crop_rotate_indexed(feat, idx, locs, oris, 4.0, crop, 0.0, 0.75)
w = torch.randn(out.shape, device=dev)
for rep in range(3):
    feat.grad = None
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    out.backward(w, retain_graph=True)
    e1.record()
    torch.cuda.synchronize()
    print(f"backward {n} crops: {e0.elapsed_time(e1):.3f} ms  (grad_out {w.numel() * 4 / 1e6:.0f} MB, grad_feat {feat.numel() * 4 / 1e6:.0f} MB)")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
out = ops.crop_rotate_indexed(feat, idx, locs, oris, 4.0, crop, 0.0, 0.75)
e1.record()
torch.cuda.synchronize()
print(f"forward  {n} crops: {e0.elapsed_time(e1):.3f} ms")
