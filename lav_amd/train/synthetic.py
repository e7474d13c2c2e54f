"""Synthetic training batches with the tensor contracts of the reference's data loaders: seeded, no
dataset on disk.  Shapes: lidar (B, Nmax, 11) + num_points (B,), heat (B,2,H,W), size / ori (B,2,H,W), bev
(B,9,H,W), ego_locs (B,T+1,2), cmds (B,), nxps (B,2), bras (B,), locs (B,N+1,T+1,2), oris (B,N+1), typs (B,N+1),
num_objs (B,)."""
from __future__ import annotations

import numpy as np
import torch

from .. import synth


def _actors(rng, B, num_objs, T):
    """Ego (index 0) + others: straight-ish futures; -y is forward in the ego frame."""
    N1 = num_objs + 1
    locs = np.zeros((B, N1, T + 1, 2), np.float32)
    oris = np.zeros((B, N1), np.float32)
    typs = np.zeros((B, N1), np.int64)
    for b in range(B):
        for n in range(N1):
            start = np.zeros(2) if n == 0 else np.array([rng.uniform(-12, 12), rng.uniform(-35, 8)])
            ori = 0.0 if n == 0 else rng.uniform(-0.6, 0.6)
            speed = rng.uniform(0.0, 0.6)
            steps = np.arange(T + 1)[:, None] * speed * np.array([[np.sin(ori), -np.cos(ori)]])
            locs[b, n] = start + steps + rng.normal(0, 0.02, (T + 1, 2))
            oris[b, n] = ori
            typs[b, n] = 1 if (n == 0 or rng.random() < 0.8) else 2
    return locs, oris, typs


def synthetic_bev_batch(B, seed=2021, num_plan=20, num_objs=6, hw=320, device="cpu"):
    """train_bev's arguments: (bev, ego_locs, cmds, nxps, bras, locs, oris, typs, num_objs)."""
    rng = np.random.default_rng(seed)
    bev = (rng.random((B, 9, hw, hw)) < 0.1).astype(np.uint8)
    locs, oris, typs = _actors(rng, B, num_objs, num_plan)
    t = lambda a: torch.from_numpy(a).to(device)
    return (t(bev), t(locs[:, 0].copy()), t(rng.integers(0, 6, B)), t(rng.uniform(-10, 10, (B, 2)).astype(np.float32)),
            t((rng.random(B) < 0.2).astype(np.int64)), t(locs), t(oris), t(typs), t(np.full(B, num_objs)))


def synthetic_lidar_batch(B, seed=2021, max_points=120000, num_plan=20, num_objs=6, hw=320, device="cpu"):
    """train_lidar's arguments: (lidars, num_points, heatmaps, sizemaps, orimaps, bev, ego_locs, cmds, nxps, bras, locs,
    oris, typs, num_objs).  Clouds: three stacked lidar-like sweeps with painted class scores and a one-hot time channel."""
    rng = np.random.default_rng(seed)
    per = max_points // 3
    lidars = np.zeros((B, max_points, 11), np.float32)
    num_points = np.zeros(B, np.int64)
    for b in range(B):
        n = int(rng.integers(max_points // 2, max_points + 1))
        pts = synth.stacked_lidar(per, seed=seed + b)[:n]
        lidars[b, : len(pts)] = pts
        num_points[b] = len(pts)
    heat = np.zeros((B, 2, hw, hw), np.float32)
    size = np.zeros((B, 2, hw, hw), np.float32)
    ori = np.zeros((B, 2, hw, hw), np.float32)
    yy, xx = np.mgrid[0:hw, 0:hw]
    for b in range(B):
        for _ in range(int(rng.integers(4, 20))):
            c, cx, cy = int(rng.integers(0, 2)), rng.uniform(20, hw - 20), rng.uniform(20, hw - 20)
            g = np.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * 3.0 ** 2)).astype(np.float32)
            heat[b, c] = np.maximum(heat[b, c], g)
            m = g > 0.3
            a = rng.uniform(-np.pi, np.pi)
            size[b, :, m] = [rng.uniform(3, 10), rng.uniform(6, 20)]
            ori[b, :, m] = [np.cos(a), np.sin(a)]
    bev_args = synthetic_bev_batch(B, seed=seed + 7, num_plan=num_plan, num_objs=num_objs, hw=hw, device=device)
    t = lambda a: torch.from_numpy(a).to(device)
    return (t(lidars), t(num_points), t(heat), t(size), t(ori)) + bev_args
