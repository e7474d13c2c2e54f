#!/usr/bin/env python3
"""Per-node cost of a linear HIP graph of N identical dependent convolutions (y = conv(y)) vs the kernel's own span."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lav_amd.ops import ConvLayer

dev = torch.device("cuda")
CASES = [("tiny 16 1x1 8x16", 1, 16, (1, 1), (0, 0), 8, 16), ("erf 64 1x3 72x64 B3", 3, 64, (1, 3), (0, 1), 72, 64),
         ("erf 128 3x1 36x32 B3", 3, 128, (3, 1), (1, 0), 36, 32), ("erf 16 3x1 144x128 B3", 3, 16, (3, 1), (1, 0), 144, 128)]
for name, B, ch, k, p, H, W in CASES:
    w = torch.randn((ch, ch, *k)) * 0.02
    layer = ConvLayer(w, padding=p, relu_pre=True, device=dev)
    x = torch.randn((B, ch, H, W), device=dev)
    bufs = [torch.empty_like(x), torch.empty_like(x)]
    layer(x, out=bufs[0])
    for n in (1, 50):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            layer(x, out=bufs[0])
            for i in range(n - 1):
                layer(bufs[i & 1], out=bufs[(i + 1) & 1])
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 40
        for _ in range(reps):
            g.replay()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps * 1e6
        print(f"{name:26s} graph of {n:3d} nodes: {dt:8.1f} us  -> {dt / n:6.2f} us/node", flush=True)
