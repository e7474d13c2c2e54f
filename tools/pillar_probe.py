#!/usr/bin/env python3
"""Pillar kernel timings (library HIP events, back-to-back launches) for several clouds + memset/copy references.
    LAV_PILLAR_TW=<w> python tools/pillar_probe.py"""
import ctypes
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import lav_amd  # noqa: E402
from lav_amd import _lib, synth  # noqa: E402

dev = torch.device("cuda:0")
lib = _lib.load()
cfg = dict(min_x=-10, max_x=70, min_y=-40, max_y=40, pixels_per_meter=4)
lm = lav_amd.LiDARModel(num_input=16, backbone="cnn", num_features=[64, 64], **cfg)
lm.load_state_dict(synth.seeded_state_dict(lm, prefix="lidar."))
ppn = lm.eval().to(dev).point_pillar_net


def read(name):
    ms, n = ctypes.c_double(), ctypes.c_int()
    lib.lav_profile_read(name.encode(), ctypes.byref(ms), ctypes.byref(n))
    return ms.value / max(n.value, 1) * 1e3


def probe(label, pts, reps=50):
    if len(sys.argv) > 1 and sys.argv[1] not in label:
        return
    pts = torch.from_numpy(pts).to(dev)
    for _ in range(3):
        ppn([pts], [len(pts)])
    torch.cuda.synchronize()
    lib.lav_profile_enable(reps + 8)
    ppn([pts], [len(pts)]); torch.cuda.synchronize(); lib.lav_profile_reset()
    for _ in range(reps):
        ppn([pts], [len(pts)])
    torch.cuda.synchronize()
    k, p = read("pointnet_scatter"), read("pillar_prep")
    lib.lav_profile_enable(0)
    nb = 4 * (len(pts) * 11 + 64 * 320 * 320)
    print(f"ZERO={os.environ.get('LAV_PILLAR_ZERO', 'rows'):4s} {label:28s} n={len(pts):7d} kernel {k:6.1f} us ({nb / k / 1e3:6.0f} GB/s, {nb / k / 8e6 * 100:4.1f}% of 8 TB/s)  prep {p:6.1f} us  stage {k + p:6.1f} us ({nb / (k + p) / 8e6 * 100:4.1f}%)", flush=True)


def ev_time(fn, reps=50):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def probe_batch(label, clouds, reps=30):
    """B clouds in one call (the kernel pair's throughput regime: the launch chain is paid once for B canvases)."""
    if len(sys.argv) > 1 and sys.argv[1] not in label:
        return
    pts = [torch.from_numpy(c).to(dev) for c in clouds]
    ns = [len(p) for p in pts]
    for _ in range(3):
        ppn(pts, ns)
    torch.cuda.synchronize()
    lib.lav_profile_enable(reps + 8)
    ppn(pts, ns); torch.cuda.synchronize(); lib.lav_profile_reset()
    for _ in range(reps):
        ppn(pts, ns)
    torch.cuda.synchronize()
    k, p = read("pointnet_scatter"), read("pillar_prep")
    lib.lav_profile_enable(0)
    nb = sum(4 * (n * 11 + 64 * 320 * 320) for n in ns)
    print(f"batch {len(pts):3d} x {label:22s} n={sum(ns):8d} kernel {k:7.1f} us ({nb / k / 1e3:6.0f} GB/s, {nb / k / 8e6 * 100:4.1f}% of 8 TB/s)  prep {p:6.1f} us  stage {k + p:7.1f} us ({nb / (k + p) / 8e6 * 100:4.1f}%)", flush=True)


rng = np.random.default_rng(0)
uni = np.concatenate([rng.uniform(-12, 72, (196608, 1)), rng.uniform(-42, 42, (196608, 1)), rng.uniform(-2.4, 1.6, (196608, 1)),
                      rng.uniform(0, 1, (196608, 8))], axis=1).astype(np.float32)
nanpts = np.full((32768, 11), np.nan, np.float32)
probe("all NaN (every tile empty)", nanpts)
probe("lidar-like 3x10923", synth.stacked_lidar(10923))
probe("lidar-like 3x65536", synth.stacked_lidar(65536))
probe("uniform 196608", uni)
probe("uniform 32768", uni[:32768])
for B in (2, 4, 8, 16, 32):
    probe_batch("lidar-like 3x10923", [synth.stacked_lidar(10923, seed=100 + b) if "seed" in synth.stacked_lidar.__code__.co_varnames else synth.stacked_lidar(10923) for b in range(B)])
if "LAV_PILLAR_TW" not in os.environ and len(sys.argv) == 1:
    canvas = torch.empty((1, 64, 320, 320), device=dev)
    src = torch.randn((1, 64, 320, 320), device=dev)
    print(f"torch zero_ 26 MB   {ev_time(lambda: canvas.zero_()):6.1f} us (back-to-back incl. launch gaps)")
    print(f"torch copy_ 26 MB   {ev_time(lambda: canvas.copy_(src)):6.1f} us")
    big = torch.empty((16, 64, 320, 320), device=dev)
    print(f"torch zero_ 419 MB  {ev_time(lambda: big.zero_(), 20):6.1f} us")
