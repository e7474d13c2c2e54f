#!/usr/bin/env python3
"""Is the MI355X train_lidar step reproducible run to run, and if not, where does it first differ?

Two trainers are built from the same seeds and driven over the same batches; after every step the loss terms and, after the
first backward, a checksum of every parameter gradient are compared.

    python tools/determinism_probe.py [steps] [--deterministic]
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lav_amd.train import LAV, TrainConfig, synthetic_lidar_batch  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 12
if "--deterministic" in sys.argv:
    torch.backends.cudnn.deterministic = True
    torch.backends.cudnn.benchmark = False
    torch.use_deterministic_algorithms(True, warn_only=True)
dev = torch.device("cuda")
batches = [synthetic_lidar_batch(2, seed=40 + i, max_points=20000, num_objs=3) for i in range(4)]


def run():
    torch.manual_seed(0)
    lav = LAV(TrainConfig(log_inference=False), dev, what="lidar")
    rows, grads = [], None
    for s in range(steps):
        torch.manual_seed(1000 + s)
        info = lav.train_lidar(*batches[s % 4])
        rows.append([v for v in info.values() if isinstance(v, float)])
        if s == 0:
            grads = {n: p.grad.detach().double().sum().item() for n, p in lav.student.named_parameters() if p.grad is not None}
    return np.array(rows), grads


a, ga = run()
b, gb = run()
print("deterministic flags:", "--deterministic" in sys.argv)
first = next((i for i in range(steps) if not np.array_equal(a[i], b[i])), None)
print("first step whose loss terms differ:", first)
for i in range(steps):
    print(f"  step {i:2d} total {a[i].sum():.6f} vs {b[i].sum():.6f}  max rel diff of the terms {np.abs(a[i] - b[i]).max() / max(np.abs(a[i]).max(), 1e-9):.2e}")
bad = [(n, ga[n], gb[n]) for n in ga if ga[n] != gb[n]]
print(f"parameter gradients after step 0: {len(bad)} of {len(ga)} checksums differ")
groups = {}
for n, x, y in bad:
    k = ".".join(n.split(".")[:3])
    groups.setdefault(k, []).append(abs(x - y) / max(abs(x), 1e-12))
for k, v in sorted(groups.items()):
    print(f"  {k:50s} {len(v):3d} tensors, max rel diff {max(v):.2e}")
