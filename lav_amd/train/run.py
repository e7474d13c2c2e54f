"""Command-line driver shared by train_bev_v2.py / train_full_v2.py and bench.py's training modes."""
from __future__ import annotations

import argparse
import json
import os
import time

import torch
import torch.distributed as dist

from .lav import LAV, TrainConfig
from .synthetic import synthetic_bev_batch, synthetic_lidar_batch


def setup_distributed():
    """(rank, world, device).  One process per GPU; RCCL ("nccl") on GPUs, gloo otherwise."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    cuda = torch.cuda.is_available()
    if cuda:
        torch.cuda.set_device(local)
    if world > 1 and not dist.is_initialized():
        dist.init_process_group("nccl" if cuda else "gloo")
    return rank, world, torch.device("cuda", local) if cuda else torch.device("cpu")


def train_loop(what, global_batch, steps, warmup, cfg=None, max_points=None, log=None):
    """Runs warmup + steps optimisation steps on a fixed synthetic shard per rank; returns (seconds for `steps`, last info)."""
    rank, world, device = setup_distributed()
    cfg = cfg or TrainConfig()
    if global_batch % world:
        raise SystemExit(f"global batch {global_batch} is not divisible by {world} ranks")
    per_rank = global_batch // world
    torch.manual_seed(cfg.seed + rank)
    lav = LAV(cfg, device, what=what)
    if what == "bev":
        batch = synthetic_bev_batch(per_rank, seed=cfg.seed + 100 * rank, device=device)
        step = lambda: lav.train_bev(*batch, other_weight=cfg.other_weight)
    else:
        batch = synthetic_lidar_batch(per_rank, seed=cfg.seed + 100 * rank, max_points=max_points or cfg.max_lidar_points, device=device)
        step = lambda: lav.train_lidar(*batch)

    def sync():
        if device.type == "cuda":
            torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    info = None
    for _ in range(warmup):
        info = step()
    sync()
    t0 = time.perf_counter()
    for i in range(steps):
        info = step()
        if log and rank == 0:
            log(i, info)
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt, info, (rank, world)


def main(what):
    ap = argparse.ArgumentParser()
    ap.add_argument("--synthetic", action="store_true", help="seeded synthetic batches (the only data source wired up)")
    ap.add_argument("--batch-size", type=int, default=64 if what == "bev" else 32, help="GLOBAL batch")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--lr", type=float, default=3e-4)
    ap.add_argument("--max-points", type=int, default=None)
    ap.add_argument("--perceive-only", action="store_true")
    ap.add_argument("--motion-only", action="store_true")
    ap.add_argument("--seed", type=int, default=2021)
    args = ap.parse_args()
    if not args.synthetic:
        raise SystemExit("only --synthetic batches are available in this build (no LMDB reader yet)")
    cfg = TrainConfig(lr=args.lr, perceive_only=args.perceive_only, motion_only=args.motion_only, seed=args.seed)
    dt, info, (rank, world) = train_loop(what, args.batch_size, args.steps, args.warmup, cfg, args.max_points,
                                         log=lambda i, inf: print(i, {k: round(v, 4) for k, v in inf.items() if isinstance(v, float)}, flush=True))
    if rank == 0:
        print(json.dumps(dict(what=what, samples_per_s=round(args.batch_size * args.steps / dt, 2), n_gpus=world,
                              global_batch=args.batch_size, steps=args.steps, s_per_step=round(dt / args.steps, 4))))
    if world > 1:
        dist.destroy_process_group()
