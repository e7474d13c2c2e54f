"""lav_crop_rotate at the frame's crop counts (1 ... 15) and at the trainers' (96): the library's own HIP-event timers around each launch
(`crop_rotate`), not a Python loop - a single crop is 19 us of launches from Python whatever the kernel takes.
    python tools/crop_probe.py"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lav_amd import _lib, ops  # noqa: E402

lib = _lib.load()
dev = torch.device("cuda")
feat = torch.randn((1, 384, 160, 160), device=dev)


def read(name):
    ms, n = ctypes.c_double(), ctypes.c_int()
    lib.lav_profile_read(name.encode(), ctypes.byref(ms), ctypes.byref(n))
    return ms.value / max(n.value, 1) * 1e3


for n in (1, 2, 4, 7, 15, 96):
    locs = torch.tensor([[3.0 * (i % 9), -5.0 - (i % 7)] for i in range(n)], device=dev)
    oris = torch.tensor([0.3 * i for i in range(n)], device=dev)
    f = feat.expand(n, -1, -1, -1) if n > 1 else feat
    for _ in range(5):
        ops.crop_rotate(feat, locs, oris, 2.0, 96, 0.0, 0.75)
    torch.cuda.synchronize()
    lib.lav_profile_enable(64)
    ops.crop_rotate(feat, locs, oris, 2.0, 96, 0.0, 0.75); torch.cuda.synchronize(); lib.lav_profile_reset()
    for _ in range(50):
        ops.crop_rotate(feat, locs, oris, 2.0, 96, 0.0, 0.75)
    torch.cuda.synchronize()
    us = read("crop_rotate")
    lib.lav_profile_enable(0)
    nb = n * 384 * 96 * 96 * 4 * 2
    print(f"crop n={n:3d}: {us:7.1f} us  ({nb / us / 1e3:6.0f} GB/s of the algorithmic 2 x output bytes, {nb / us / 8e6 * 100:4.1f} % of 8 TB/s)", flush=True)
