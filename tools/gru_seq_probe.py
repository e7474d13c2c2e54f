#!/usr/bin/env python3
"""Forward + backward time of the planners' training GRUs: liblav_amd's sequence GRU vs torch's nn.GRU (MIOpen RNN)."""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from lav_amd import ops  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for name, R, T, I, H in (("plan (6 x 32 rows)", 192, 20, 4, 512), ("cast (32 + 96 rows)", 128, 20, 512, 384), ("plan (6 x 4 rows)", 24, 20, 4, 512)):
    gru = torch.nn.GRU(I, H, batch_first=True).to(dev)
    u = torch.randn((R, T, I), device=dev, requires_grad=True)
    h0 = torch.randn((R, H), device=dev, requires_grad=True)
    w = torch.randn((R, T, H), device=dev)

    def run_torch():
        out, _ = gru(u, h0[None])
        (out * w).sum().backward()

    def run_hip():
        x = torch.nn.functional.linear(u, gru.weight_ih_l0, gru.bias_ih_l0)
        out = ops.gru_seq(x, h0, gru.weight_hh_l0, gru.bias_hh_l0, T)
        (out * w).sum().backward()

    print(f"{name:22s} R={R:4d} H={H}: nn.GRU fwd+bwd {timeit(run_torch):7.3f} ms   lav gru_seq fwd+bwd {timeit(run_hip):7.3f} ms", flush=True)
