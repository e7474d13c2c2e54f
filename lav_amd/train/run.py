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
    """(rank, world, device).  One process per GPU; RCCL ("nccl") on GPUs, gloo otherwise.  LAV_DIST_BACKEND=gloo forces
    gloo with HIP tensors (gradients hop through the host) and lets ranks share a GPU (local rank modulo the device count):
    how the data-parallel step over the HIP autograd functions is exercised on a one-GPU box (tests/test_gpu_train.py)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    cuda = torch.cuda.is_available()
    backend = os.environ.get("LAV_DIST_BACKEND") or ("nccl" if cuda else "gloo")
    if cuda:
        if backend == "gloo":
            local %= torch.cuda.device_count()
        torch.cuda.set_device(local)
    if world > 1 and not dist.is_initialized():
        dist.init_process_group(backend)
    return rank, world, torch.device("cuda", local) if cuda else torch.device("cpu")


def train_loop(what, global_batch, steps, warmup, cfg=None, max_points=None, log=None, profile_steps=0, device=None, wrap=None):
    """Runs warmup + steps optimisation steps on a fixed synthetic shard per rank; returns (seconds for `steps`, last info).
    profile_steps > 0: that many EXTRA steps after the timed region with the library's HIP-event timers armed; their per-kernel
    times and the algorithmic work lav_amd.ops counted over the same steps come back in info["hand_kernels"]."""
    rank, world, dev0 = setup_distributed()
    device = device if device is not None else dev0    # device=cpu: the same trainer on torch CPU ops (bench.py's cpu_baseline)
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

    if wrap is not None:      # a context manager factory over the trainer (bench.py's CPU leg swaps the teacher's kernels for torch ops)
        with wrap(lav):
            return _run_steps(step, steps, warmup, device, world, rank, log) + ((rank, world),)
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
    if profile_steps > 0 and device.type == "cuda":   # every rank steps (the gradient exchange needs them all)
        import ctypes
        from .. import _lib, ops
        lib = _lib.load()
        lib.lav_profile_enable(4096)
        lib.lav_profile_reset()
        for k in ops.train_work:
            ops.train_work[k] = 0
        for _ in range(profile_steps):
            step()
        torch.cuda.synchronize()
        kernels = {}
        for name in ("crop_rotate_backward", "crop_rotate", "gru_seq_forward", "gru_seq_backward", "gru_plan", "scatter_max", "pillar_decorate",
                     "bn_train_fwd", "bn_train_bwd", "conv_wgrad", "conv2d"):
            ms, n = ctypes.c_double(), ctypes.c_int()
            lib.lav_profile_read(name.encode(), ctypes.byref(ms), ctypes.byref(n))
            if n.value:
                kernels[name] = dict(calls_per_step=n.value / profile_steps, ms_per_call=ms.value / n.value, ms_per_step=ms.value / profile_steps)
        lib.lav_profile_enable(0)
        info = dict(info, hand_kernels=dict(kernels=kernels, work_per_step={k: v / profile_steps for k, v in ops.train_work.items()}))
    if world > 1:
        dist.barrier()
    return dt, info, (rank, world)


def _run_steps(step, steps, warmup, device, world, rank, log):
    info = None
    for _ in range(warmup):
        info = step()
    t0 = time.perf_counter()
    for i in range(steps):
        info = step()
        if log and rank == 0:
            log(i, info)
    return time.perf_counter() - t0, info


def load_config(path, **overrides) -> TrainConfig:
    """TrainConfig from the reference's YAML (config_v2.yaml): every key that TrainConfig has is taken from the file,
    the others (data paths, controller gains ...) are not part of the training step.  `distill` is read by
    lav_final_v2.py:244 but absent from config_v2.yaml (team_code_v2/config.yaml:11 says True): injected as True."""
    import dataclasses
    import yaml
    fields = {f.name for f in dataclasses.fields(TrainConfig)}
    vals = {}
    if path:
        with open(path, "r") as f:
            raw = yaml.safe_load(f) or {}
        vals = {k: v for k, v in raw.items() if k in fields}
    vals.setdefault("distill", True)
    vals.update({k: v for k, v in overrides.items() if v is not None})
    return TrainConfig(**vals)


def resolve_checkpoints(what, args):
    """Which checkpoints a run starts from.  Command-line paths win; otherwise a run on recorded routes follows the reference's
    rules for the *_model_dir keys of config_v2.yaml (lav/lav_final_v2.py:42-72): the privileged teacher `bev_model_dir` is
    ALWAYS loaded by train_full_v2, `lidar_model_dir` unless --perceive-only, `uniplanner_dir` unless --perceive-only or
    --motion-only; train_bev_v2 starts from scratch.  A missing file is an error there, never a silent fall back to seeded random
    weights (that is what --synthetic runs use, explicitly)."""
    paths = dict(lidar=args.lidar, bev=args.bev, uniplanner=args.uniplanner)
    if args.synthetic or not args.config_path or what != "lidar":
        return paths
    import yaml
    with open(args.config_path, "r") as f:
        raw = yaml.safe_load(f) or {}
    wanted = dict(bev="bev_model_dir")
    if not args.perceive_only:
        wanted["lidar"] = "lidar_model_dir"
        if not args.motion_only:
            wanted["uniplanner"] = "uniplanner_dir"
    for name, key in wanted.items():
        if paths[name]:
            continue
        rel = raw.get(key)
        if not rel:
            raise SystemExit(f"{args.config_path} has no `{key}` and --{name} was not given: train_full_v2 loads it (lav/lav_final_v2.py:42-72)")
        cands = [rel, os.path.join(os.path.dirname(os.path.abspath(args.config_path)), rel)]
        hit = next((c for c in cands if os.path.isfile(c)), None)
        if hit is None:
            raise SystemExit(f"checkpoint `{key}: {rel}` of {args.config_path} not found (tried {cands}); pass --{name} PATH, or --synthetic "
                             "to train seeded random weights on synthetic batches")
        paths[name] = hit
    return paths


def set_deterministic(on: bool = True):
    """Run-to-run bit reproducibility of a training step.  liblav_amd's kernels (pillar front end, scatter-max, crop gather,
    GRU tape, BatchNorm) reduce in a fixed order; what differs between two runs of the default configuration is torch's own
    atomics-based backward kernels and MIOpen's algorithm choice (tools/determinism_probe.py: already the forward loss of step 0
    differs by 3e-7, 181 of 189 parameter gradients after it).  With these switches two runs agree bit for bit."""
    torch.backends.cudnn.deterministic = bool(on)
    if on:
        torch.backends.cudnn.benchmark = False
    # Round 6: reproducible from PROCESS to process too.  With benchmark off torch still lets MIOpen "find" - time the applicable solvers and
    # keep the fastest -, so which (deterministic) solver a layer gets is a measurement, and two processes on one box disagreed about once in
    # three starts (tools/determinism_xproc.py: the first difference at step 0, 1 or 2; inside a process 24 of 24 repeats of step 0 agree).
    # MIOpen's immediate mode takes the solver from its database / heuristic instead of a timing.
    mi = getattr(torch.backends, "miopen", None)
    if mi is not None and hasattr(mi, "immediate"):
        mi.immediate = bool(on)
    torch.use_deterministic_algorithms(bool(on), warn_only=True)


def other_weight_schedule(it, beta=0.8):
    """lav/train_bev_v2.py:38-39"""
    return 1 - beta ** (it / 4000)


def main(what):
    """Command line of lav/train_full_v2.py:48-70 / lav/train_bev_v2.py:42-63 (same flags and defaults): the recorded routes
    under the config's `data_dir` are read by lav_amd.data ('temporal_lidar_painted' / 'temporal_bev' loaders, every rank its
    own shard of each epoch).  What this build adds: --synthetic / --steps-per-epoch (seeded synthetic batches instead of a
    data set), --save-dir, --lidar / --bev / --uniplanner (checkpoints to start from), --max-points, --log-every."""
    ap = argparse.ArgumentParser()
    ap.add_argument("--config-path", default=None, help="the reference's config_v2.yaml (training keys are read from it)")
    ap.add_argument("--device", default="cuda", choices=["cuda", "cpu"])
    ap.add_argument("--perceive-only", action="store_true")
    ap.add_argument("--motion-only", action="store_true")
    ap.add_argument("--num-epoch", type=int, default=64 if what == "lidar" else 160)
    ap.add_argument("--num-per-log", type=int, default=100, help="log per iter")
    ap.add_argument("--num-per-save", type=int, default=1, help="save per epoch")
    ap.add_argument("--batch-size", type=int, default=32 if what == "lidar" else 256, help="GLOBAL batch, split over the ranks")
    ap.add_argument("--lr", type=float, default=3e-4)
    ap.add_argument("--weight-decay", type=float, default=2e-4, help="accepted for command-line compatibility; the reference never passes it to Adam")
    ap.add_argument("--num-workers", type=int, default=16, help="DataLoader workers (recorded routes only)")
    ap.add_argument("--seed", type=int, default=2021)
    ap.add_argument("--synthetic", action="store_true", help="seeded synthetic batches instead of the config's data_dir")
    ap.add_argument("--steps-per-epoch", type=int, default=20, help="iterations that make one epoch of synthetic data")
    ap.add_argument("--save-dir", default="checkpoints")
    ap.add_argument("--lidar", default=None, help="lidar_*.th to start from")
    ap.add_argument("--bev", default=None, help="bev_*.th to start from / the teacher of train_full_v2")
    ap.add_argument("--uniplanner", default=None, help="uniplanner_*.th to start from")
    ap.add_argument("--max-points", type=int, default=None)
    ap.add_argument("--log-every", type=int, default=None, help="steps between the eval-mode log inference (default: --num-per-log)")
    ap.add_argument("--deterministic", action="store_true", help="bit-reproducible steps: deterministic torch / MIOpen algorithms "
                    "(liblav_amd's own kernels always are); slower convolution gradients")
    args = ap.parse_args()
    if args.deterministic:
        set_deterministic(True)
    if not args.synthetic and not args.config_path:
        raise SystemExit("recorded routes are read from the data_dir of --config-path (or pass --synthetic)")
    rank, world, device = setup_distributed()
    if args.device == "cpu":
        device = torch.device("cpu")
    cfg = load_config(args.config_path, lr=args.lr, perceive_only=args.perceive_only, motion_only=args.motion_only, seed=args.seed,
                      log_every=args.log_every if args.log_every is not None else args.num_per_log)
    if args.batch_size % world:
        raise SystemExit(f"global batch {args.batch_size} is not divisible by {world} ranks")
    per_rank = args.batch_size // world
    paths = resolve_checkpoints(what, args)
    ck = {k: torch.load(v, map_location="cpu") for k, v in paths.items() if v}
    torch.manual_seed(cfg.seed + rank)
    lav = LAV(cfg, device, what=what, checkpoints=ck)
    log = lambda it, inf: print(it, {k: round(v, 4) for k, v in inf.items() if isinstance(v, float)}, flush=True)
    loader = None
    if not args.synthetic:
        from ..data import get_data_loader
        loader = get_data_loader("temporal_bev" if what == "bev" else "temporal_lidar_painted", args, rank=rank, world=world)
        if len(loader) == 0:
            raise SystemExit(f"{args.config_path}: data_dir holds fewer frames than one batch of {args.batch_size}")

    def batches(epoch):
        if loader is not None:
            if world > 1:
                loader.sampler.set_epoch(epoch)
            yield from loader
            return
        for it in range(args.steps_per_epoch):
            seed = cfg.seed + 1000003 * epoch + 1009 * it + 100 * rank
            if what == "bev":
                yield synthetic_bev_batch(per_rank, seed=seed, device=device)
            else:
                yield synthetic_lidar_batch(per_rank, seed=seed, max_points=args.max_points or cfg.max_lidar_points, device=device)

    global_it, t0 = 0, time.perf_counter()
    for epoch in range(args.num_epoch):
        for batch in batches(epoch):
            if what == "bev":
                info = lav.train_bev(*batch, other_weight=other_weight_schedule(global_it))
            else:
                info = lav.train_lidar(*batch)
            if global_it % args.num_per_log == 0 and rank == 0:
                log(global_it, info)
            global_it += 1
        (lav.bev_scheduler if what == "bev" else lav.lidar_scheduler).step()       # once per epoch (train_full_v2.py:33)
        if (epoch + 1) % args.num_per_save == 0 and rank == 0:
            os.makedirs(args.save_dir, exist_ok=True)
            for name in (("bev",) if what == "bev" else ("lidar", "uniplanner")):
                path = os.path.join(args.save_dir, f"{name}_{epoch + 1}.th")
                torch.save(lav.state_dict(name), path)
                print(f"saved to {path}", flush=True)
    if device.type == "cuda":
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    in_sync = None
    if world > 1:   # data parallel keeps the replicas identical: compare a checksum of every trained parameter across the ranks
        # (BatchNorm's running statistics are per-rank by design - DDP re-broadcasts rank 0's before each forward - and are left out)
        mods = (lav.bev_planner,) if what == "bev" else (lav.lidar_model, lav.uniplanner)
        mine = torch.stack([p_.detach().double().sum() for m in mods for p_ in m.parameters() if p_.requires_grad]).to(device)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        in_sync = all(torch.equal(every[0], e) for e in every[1:])
    if rank == 0:
        print(json.dumps(dict(what=what, samples_per_s=round(args.batch_size * global_it / dt, 2), n_gpus=world, replicas_in_sync=in_sync,
                              global_batch=args.batch_size, steps=global_it, epochs=args.num_epoch,
                              data="synthetic batches" if loader is None else f"{len(loader.dataset)} recorded frames",
                              lr=(lav.bev_optim if what == "bev" else lav.lidar_optim).param_groups[0]["lr"],
                              scheduler_epochs=(lav.bev_scheduler if what == "bev" else lav.lidar_scheduler).last_epoch)))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
