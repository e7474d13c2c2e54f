#!/usr/bin/env python3
"""Benchmark of the LAV full-agent forward on MI355X (BASELINE.json metric: frames/s, 32k-point LiDAR + 3 cams).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   (N > 1: one replica per GPU)

One step = one pass of lav_amd.frame.FramePipeline.step - everything LAVAgent.run_step does on the GPU for
one 20 Hz tick (half-sweep concat, ego-box removal, ERFNet + softmax on 3 cameras, point painting, 3-sweep
temporal stacking, pillar scatter, BEV backbone + heads, detection decode, ResNet-18 embedding of the ego crop
(+ one crop per detected vehicle), cast + plan GRUs, brake net) - on synthetic inputs already resident in HBM.
Inference is a closed loop with per-vehicle state: it does not shard, so N > 1 runs N independent replicas
("replicas only") and reports the aggregate frames/s.

Prints ONE JSON line (rank 0) with `roofline` for the pillar point-net/scatter kernel (HIP events recorded by
the library on its launch stream) and `cpu_baseline` (the oracle's full frame on the host cores, N=1 only).
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
MFMA_F32_PEAK_TFLOPS = 157.3
MFMA_BF16_PEAK_TFLOPS = 2500.0   # dense bf16 MFMA (same guide); the split convolution kernels run on these pipes
DTYPE = ("f32 (convolution contractions on the matrix cores with f32 accumulate at the error of an fp32 dot product, DESIGN 4.3: the layers on "
         "the split-operand kernel - BEV backbone, heads, crop stems, the brake net's mid-size layers - and ERFNet's persistent runs of pairs as two "
         "power-of-two-scaled fp16 pieces per fp32 operand and three partial products, the small layers and the backbone's 1x1 / 4x4-stride-4 "
         "up-convolutions on the fp32 MFMA (exact products); LAV_INFER_PRECISION=bf16x6 restores round 5's three bf16 pieces / six products "
         "everywhere; LAV_CONV_PRECISION=f32 selects the fp32-MFMA kernels)")


def build_pipeline(device, eager=False):
    import lav_amd
    from lav_amd import synth
    from lav_amd.frame import FramePipeline, GraphedFramePipeline
    from lav_amd.rgb import RGBBrakePredictionModel, RGBSegmentationModel
    cfg = dict(min_x=-10, max_x=70, min_y=-40, max_y=40, pixels_per_meter=4)
    y_off = 1 + cfg["min_x"] / ((cfg["max_x"] - cfg["min_x"]) / 2)
    lm = lav_amd.LiDARModel(num_input=16, backbone="cnn", num_features=[64, 64], **cfg)
    bp = lav_amd.BEVPlanner(pixels_per_meter=4, crop_size=96, feature_x_jitter=1.5, feature_angle_jitter=20, x_offset=0,
                            y_offset=y_off, num_cmds=6, num_plan=20, num_plan_iter=5, num_frame_stack=2)
    up = lav_amd.UniPlanner(bp, pixels_per_meter=4, crop_size=96, feature_x_jitter=1.5, feature_angle_jitter=20,
                            x_offset=0, y_offset=y_off, num_cmds=6, num_plan=20, num_input_feature=384, num_plan_iter=5)
    seg = RGBSegmentationModel([4, 6, 7, 10])
    bra = RGBBrakePredictionModel([4, 6, 7, 10])
    sds = dict(lidar=synth.seeded_state_dict(lm, prefix="lidar."), uni=synth.seeded_state_dict(up, prefix="uni."),
               seg=synth.seeded_state_dict(seg, prefix="seg."), bra=synth.seeded_state_dict(bra, prefix="bra."))
    lm.load_state_dict(sds["lidar"]); up.load_state_dict(sds["uni"]); seg.load_state_dict(sds["seg"]); bra.load_state_dict(sds["bra"])
    for m in (lm, up, seg, bra):
        m.eval().to(device)
    cls = FramePipeline if eager else GraphedFramePipeline
    return cls(lm, up, seg, bra, 1.5, 2.4, num_frame_stack=2, device=device), sds, (lm, up, seg, bra)


def synthetic_inputs(device, n_ticks=4, n_points=32768):
    from lav_amd import synth
    ticks = [synth.lidar_sweep(n_points, name=f"tick{i}") for i in range(n_ticks)]
    cams, tel = synth.rgb_frames()
    rgbs = [c[..., :3][..., ::-1] for c in cams]                                 # BGRA -> RGB (lav_agent_fast.py:252-254)
    all_rgb = np.stack(rgbs, 0).transpose(0, 3, 1, 2).astype(np.float32)          # (3,3,288,256)
    wide = np.concatenate(rgbs, axis=1)[None].transpose(0, 3, 1, 2).astype(np.float32)   # (1,3,288,768)
    tel_rgb = tel[..., :3][..., ::-1][:-96][None].transpose(0, 3, 1, 2).astype(np.float32)  # (1,3,192,480)
    host = dict(ticks=ticks, all_rgbs=all_rgb, rgbs=wide, tel_rgbs=tel_rgb, nxp=np.array([0.0, -10.0], np.float32))
    dev = dict(ticks=[torch.from_numpy(t).to(device) for t in ticks], all_rgbs=torch.from_numpy(all_rgb).to(device),
               rgbs=torch.from_numpy(wide).to(device), tel_rgbs=torch.from_numpy(tel_rgb).to(device),
               nxp=torch.from_numpy(host["nxp"]).to(device))
    return host, dev


def pose(i):
    """EKF pose stream: gentle forward motion, 0.05 rad / 5 frames of yaw."""
    return np.array([0.2 * i, 0.02 * i]), 0.01 * i


def cpu_baseline(sds, host, budget_s=20.0):
    """The oracle's full frame on the host cores (kind "port"), bounded to ~budget_s of CPU work."""
    from lav_amd.rgb import RGBBrakePredictionModel, RGBSegmentationModel
    from oracle import frame as oframe
    torch.set_num_threads(min(16, os.cpu_count() or 1))   # more threads only oversubscribe these small layers
    seg = RGBSegmentationModel([4, 6, 7, 10]).eval(); seg.load_state_dict(sds["seg"])
    bra = RGBBrakePredictionModel([4, 6, 7, 10]).eval(); bra.load_state_dict(sds["bra"])
    lsd = {k: v for k, v in sds["lidar"].items()}
    usd = {k: v for k, v in sds["uni"].items()}
    pn = {k[len("point_pillar_net."):]: v.numpy() for k, v in lsd.items() if k.startswith("point_pillar_net.")}
    hist = dict(lidars=[], locs=[], oris=[])
    t_args = dict(all_rgbs=torch.from_numpy(host["all_rgbs"]), rgbs=torch.from_numpy(host["rgbs"]),
                  tel_rgbs=torch.from_numpy(host["tel_rgbs"]))
    # fill the 15-frame history cheaply with painted sweeps (not timed)
    from oracle import paint as opaint
    for i in range(14):
        loc, ori = pose(i)
        cur = opaint.preprocess(np.concatenate([host["ticks"][i % len(host["ticks"])], host["ticks"][(i + 1) % len(host["ticks"])]]))
        hist["lidars"].append(np.concatenate([cur, np.zeros((len(cur), 4), np.float32)], axis=1))
        hist["locs"].append(loc); hist["oris"].append(ori)
    times = []
    i = 14
    t_start = time.time()
    while True:
        loc, ori = pose(i)
        t0 = time.time()
        oframe.frame(host["ticks"][i % len(host["ticks"])], host["ticks"][(i - 1) % len(host["ticks"])], hist,
                     t_args["all_rgbs"], t_args["rgbs"], t_args["tel_rgbs"], seg, bra, lsd, usd, pn,
                     torch.from_numpy(host["nxp"]), 3, loc, ori)
        times.append(time.time() - t0)
        i += 1
        if time.time() - t_start > budget_s or len(times) >= 12:
            break
    med = float(np.median(times[1:] if len(times) > 1 else times))
    return dict(value=1.0 / med, unit="frames/s", cores=torch.get_num_threads(), kind="port",
                sample=f"{len(times)} full frames of the oracle (numpy pillar/paint + torch-CPU conv/GRU), median of all but the first; 32k-pt ticks, 3 cams")


def train_bench(args):
    """BASELINE.json metric (ii): samples/s of train_full_v2 (config #5: batch 32) / train_bev_v2 (config #4: batch 64) on
    synthetic batches, one process per GPU, gradient all-reduce over RCCL.  The GLOBAL batch is BASELINE's at every N
    (strong scaling: 32 / N resp. 64 / N samples per GPU; `--batch B` fixes the per-GPU batch instead = weak scaling).
    One step = forward + backward + Adam, plus - for train_full - the reference's eval-mode inference of sample 0 that
    feeds its log (lav_final_v2.py:228-236) every `--log-every` steps: 1 = the reference's cadence (it runs it on every
    step and logs every 100th), 100 = only on the steps that are logged.  Convolutions and Linear layers run on torch autograd
    (MIOpen / rocBLAS); BatchNorm + ReLU (+ residual), the pillar front end, scatter-max, the crops and the GRU recurrences are
    liblav_amd kernels."""
    # MIOpen's on-disk user database can hold solver choices that another process made under other rules (a test session in torch's
    # deterministic mode leaves the naive weight-gradient kernel for these very shapes: 634 instead of 128 ms per step, round 5): the
    # training bench searches for itself, in a directory of its own (the search runs inside the warm-up steps)
    if "MIOPEN_USER_DB_PATH" not in os.environ:
        _d = os.path.join(os.path.expanduser("~"), ".cache", "lav_amd", "miopen_bench")   # (stable: nothing piles up under /tmp)
        try:
            os.makedirs(_d, exist_ok=True)
        except OSError:
            import tempfile
            _d = os.path.join(tempfile.gettempdir(), "lav_amd_miopen_bench")
            os.makedirs(_d, exist_ok=True)
        os.environ["MIOPEN_USER_DB_PATH"] = _d
        os.environ.setdefault("MIOPEN_CUSTOM_CACHE_DIR", _d)
    from lav_amd.train import TrainConfig
    from lav_amd.train.run import train_loop
    what = "lidar" if args.mode == "train_full" else "bev"
    world = int(os.environ.get("WORLD_SIZE", "1"))
    base = 32 if what == "lidar" else 64
    if args.batch:
        per, scaling = args.batch, "weak"
    else:
        if base % world:
            raise SystemExit(f"BASELINE's global batch {base} does not divide over {world} ranks (use --batch)")
        per, scaling = base // world, "strong"
    steps = args.steps if args.steps != 100 else 10
    warmup = min(args.warmup, 3)
    cfg = TrainConfig(log_every=args.log_every)
    dt, info, (rank, world) = train_loop(what, per * world, steps, warmup, cfg=cfg, profile_steps=2)
    roofline = None
    hk = info.get("hand_kernels")
    if rank == 0 and hk and "crop_rotate_backward" in hk["kernels"]:
        # The hand-written kernel that takes the most time in a training step: the gradient of the rotated crops w.r.t. the
        # feature maps (gather form).  HBM-bound: every output-pixel gradient is read once, every map pixel written
        # once; HIP events of the library's launch timers over two extra steps after the timed region.
        k = hk["kernels"]["crop_rotate_backward"]
        nbytes = hk["work_per_step"]["crop_rotate_backward_bytes"] / max(hk["work_per_step"]["crop_rotate_backward_calls"], 1)
        gbs = nbytes / (k["ms_per_call"] * 1e-3) / 1e9
        roofline = dict(bound="hbm", kernel="k_crop_rotate_bwd (gradient of the rotated crops, gather form)", achieved=round(gbs, 1), peak=HBM_PEAK_GBS,
                        unit="GB/s", frac=round(gbs / HBM_PEAK_GBS, 4), traffic=None, algorithmic_bytes=int(nbytes), avg_kernel_us=round(k["ms_per_call"] * 1e3, 1),
                        launches=int(round(k["calls_per_step"] * 2)),
                        hand_kernels_ms_per_step={n: round(v["ms_per_step"], 3) for n, v in hk["kernels"].items()},
                        gru_seq_tflops={d: round(hk["work_per_step"][f"gru_seq_{d}_flops"] / (hk["kernels"][f"gru_seq_{d}"]["ms_per_step"] * 1e-3) / 1e12, 2)
                                        for d in ("forward", "backward") if f"gru_seq_{d}" in hk["kernels"]},
                        note="the convolutions of the step run on MIOpen (~55 % of its GPU time, profiles/r03_train_full_kernel_top.txt; BatchNorm + ReLU are the fused lav_bn_train_* pairs): this is the roofline of the largest HAND-WRITTEN kernel of the step, not of the step")
    elif rank == 0 and hk and any(n.startswith("gru_seq") for n in hk["kernels"]):
        # no crop gradient in this step: the sequence GRU (recurrent GEMM on MFMA fused with the gates, one launch per time step)
        name = max((n for n in hk["kernels"] if n.startswith("gru_seq")), key=lambda n: hk["kernels"][n]["ms_per_step"])
        k = hk["kernels"][name]
        tf = hk["work_per_step"][f"{name}_flops"] / (k["ms_per_step"] * 1e-3) / 1e12
        roofline = dict(bound="mfma", kernel=f"lav_{name} (k_gru_fwd_step / k_gru_bwd_step, one launch per time step)", achieved=round(tf, 2),
                        peak=MFMA_F32_PEAK_TFLOPS, unit="TFLOP/s", frac=round(tf / MFMA_F32_PEAK_TFLOPS, 4), traffic=None,
                        algorithmic_flops=hk["work_per_step"][f"{name}_flops"] / max(k["calls_per_step"], 1e-9), avg_kernel_us=round(k["ms_per_call"] * 1e3, 1),
                        launches=int(round(k["calls_per_step"] * 2)),
                        hand_kernels_ms_per_step={n: round(v["ms_per_step"], 3) for n, v in hk["kernels"].items()},
                        note="a step of these GRUs is a 192..400 x 512 x 1536 GEMM: latency-bound, not matrix-bound; the convolutions of the "
                             "step run on MIOpen (BatchNorm + ReLU: the fused lav_bn_train_* pairs): this is the roofline of the largest HAND-WRITTEN kernel of the step, not of the step")
    if rank == 0 and roofline is not None and hk and "conv_wgrad" in hk["kernels"] and hk["work_per_step"].get("conv_wgrad_flops"):
        # the 3x3 / 7x7 weight gradients on lav_conv_wgrad: round 5 bf16x6 (six bf16 matrix products per fp32 product), round 6 - train_full's
        # default - f16x3 (three fp16 ones): the executed rate against the dense 16-bit peak is 6x / 3x the algorithmic one
        from lav_amd.train import hipnn
        from lav_amd import _lib as _l
        with hipnn.use_precision(cfg.conv_precision or "f16x3"):
            nprod = 3 if hipnn.train_precision() == _l.CONV_F16X3 else 6
        k = hk["kernels"]["conv_wgrad"]
        tf = hk["work_per_step"]["conv_wgrad_flops"] / (k["ms_per_step"] * 1e-3) / 1e12
        roofline["conv_wgrad"] = dict(bound="mfma", achieved=round(nprod * tf, 1), peak=2500.0, unit=f"TFLOP/s (16-bit MFMA, executed = {nprod} x algorithmic)",
                                      frac=round(nprod * tf / 2500.0, 4), executed_per_algorithmic=nprod,
                                      fp32_equivalent_tflops=round(tf, 1), ms_per_step=round(k["ms_per_step"], 3), launches_per_step=round(k["calls_per_step"], 1))
        roofline["note"] = ("the 3x3 / 7x7 convolutions' forward and data gradients are lav_conv2d launches, their weight gradients lav_conv_wgrad (round 6: "
                            "all three on two fp16 pieces per operand, the scales from the bounds the fused BatchNorm launches leave); transposed / 1x1 / small-map "
                            "weight-gradient convolutions stay on MIOpen")
    if rank == 0 and roofline is not None and hk and "bn_train_fwd" in hk["kernels"]:
        # the fused train-mode BatchNorm + ReLU (+ residual) pairs (lav_bn_train_*): HBM bound, algorithmic bytes = passes over the activation
        for d in ("fwd", "bwd"):
            k = hk["kernels"][f"bn_train_{d}"]
            gbs = hk["work_per_step"][f"bn_train_{d}_bytes"] / (k["ms_per_step"] * 1e-3) / 1e9
            roofline[f"bn_train_{d}"] = dict(bound="hbm", achieved=round(gbs, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(gbs / HBM_PEAK_GBS, 4),
                                             ms_per_step=round(k["ms_per_step"], 3), launch_pairs_per_step=round(k["calls_per_step"], 1))
    if rank == 0:
        print(json.dumps(dict(metric=f"samples/s {args.mode}_v2 (synthetic batch)", value=round(per * world * steps / dt, 2),
                              unit="samples/s", n_gpus=world, steps=steps, warmup=warmup, ms_per_step=round(dt / steps * 1e3, 2),
                              higher_is_better=True, scaling=scaling, vs_baseline=None, dtype=("f32 (3x3 / 7x7 convolutions forward / data gradient / weight gradient: liblav_amd on the 16-bit matrix cores with fp32 operands split into "
                                     + ("two scaled fp16 pieces, three products" if not os.environ.get("LAV_TRAIN_PRECISION", "").startswith("bf16") else "three bf16 pieces, six products")
                                     + ", f32 accumulate; transposed / 1x1 convolutions: MIOpen f32; BatchNorm+ReLU, pillar, crop, GRU: liblav_amd f32)"), data="synthetic",
                              config=dict(workload=f"{args.mode}_v2 step: fwd + bwd + Adam, per-GPU batch {per}, global batch {per * world}"
                                          + (f", 120000-point clouds, 320x320 maps; log-only eval inference every {args.log_every} step(s)"
                                             if what == "lidar" else ", (9,320,320) BEV"),
                                          parallelism=f"dp{world}", global_batch=per * world, log_every=args.log_every,
                                          loss=round(info["loss"], 4)),
                              roofline=roofline,
                              cpu_baseline=cpu_train_baseline(what) if (args.cpu_train_baseline and world == 1) else None)))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


def training_lines(with_cpu=True, world=1, rank=0):
    """BASELINE metric (ii) inside the default run: train_full_v2 (global batch 32) and train_bev_v2 (global batch 64), 3 warm-up +
    5 timed steps each at the reference's cadence (--log-every 1), as child processes (a fresh CUDA context each: the
    training graph's allocator state must not sit beside the frame's HIP graphs), each with its roofline and a bounded
    cpu_baseline.
    world > 1 (round 6: `bench.py --gpus N` under torch.distributed.run, the driver's scaling runs): EVERY rank calls this and
    starts its own child with its RANK / LOCAL_RANK / WORLD_SIZE; the N children form a process group of their own on
    MASTER_PORT + 1 / + 2 and run the data-parallel step at BASELINE's GLOBAL batch (32 / N resp. 64 / N samples per GPU, one
    bucketed RCCL all-reduce per step) - `training.train_full.value` over the driver's N = 1, 2, 4, 8 lines is the scaling curve of
    north_star's ">= 6x at 8 GPUs".  Rank 0's child prints the line."""
    import subprocess
    out = {}
    for k, mode in enumerate(("train_full", "train_bev")):
        cmd = [sys.executable, os.path.abspath(__file__), "--mode", mode, "--steps", "5", "--warmup", "3", "--log-every", "1"]
        if with_cpu and world == 1:
            cmd.append("--cpu-train-baseline")
        env = dict(os.environ)
        if world > 1:
            env["MASTER_PORT"] = str(int(os.environ.get("MASTER_PORT", "29500")) + 1 + k)
            # (torch.distributed.run tells its workers to use the AGENT's store at MASTER_PORT; the children rendezvous on a port of
            # their own, where rank 0 must host the store itself - with the agent's variables inherited they wait for ever)
            for name in [n for n in env if n.startswith("TORCHELASTIC_")]:
                env.pop(name)
            d = os.path.join(os.path.expanduser("~"), ".cache", "lav_amd", f"miopen_bench_rank{rank}")   # (ranks do not share a find database)
            try:
                os.makedirs(d, exist_ok=True)
                env.setdefault("MIOPEN_USER_DB_PATH", d); env.setdefault("MIOPEN_CUSTOM_CACHE_DIR", d)
            except OSError:
                pass
        try:
            r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                               timeout=float(os.environ.get("LAV_BENCH_TRAIN_TIMEOUT", "420")))
            if rank == 0:
                line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
                out[mode] = json.loads(line)
        except Exception as e:   # a failed training line must not take the inference line with it
            out[mode] = dict(error=f"{type(e).__name__}: {e}"[:300])
    return out


def cpu_train_baseline(what, budget_batch=16, steps=6):
    """The same optimisation step on the host cores (torch CPU ops of the same modules: the port of the reference's trainer),
    one warm-up + `steps` timed steps at a reduced batch (about 10 s of CPU work): a bounded sample for orientation, not a target."""
    from lav_amd.train import TrainConfig
    from lav_amd.train.run import train_loop
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    t0 = time.time()
    restore = None
    if what == "lidar":
        # The product's train-mode PointPillar front end does its index work with liblav_amd and has no CPU path; for THIS leg the
        # oracle's restatement of it (numpy index work + the module's own PointNet layers + index_reduce) stands in - the rest of
        # the step (backbone, heads, planners, losses, Adam) is the same trainer code on torch CPU ops, log inference off (it
        # runs on the HIP inference kernels), and the frozen teacher is evaluated through its modules' torch code paths
        # (oracle/train_cpu.py).  A reduced sample: batch 4, 20 000-point clouds.
        from lav_amd.point_pillar import PointPillarNet
        from oracle import train_cpu
        budget_batch, steps = 4, 3
        restore = PointPillarNet.forward_train
        PointPillarNet.forward_train = lambda self, lidars, num_points: train_cpu.pillar_forward_train(self, lidars, num_points)
    try:
        dt, info, _ = train_loop(what, budget_batch, steps, 1, cfg=TrainConfig(log_every=100, log_inference=False), device=torch.device("cpu"),
                                 max_points=20000 if what == "lidar" else None,
                                 wrap=(lambda lav: train_cpu.teacher_on_cpu(lav.bev_planner)) if what == "lidar" else None)
    finally:
        if restore is not None:
            PointPillarNet.forward_train = restore
    return dict(value=round(budget_batch * steps / dt, 3), unit="samples/s", cores=torch.get_num_threads(), kind="port",
                sample=f"{steps} steps (after 1 warm-up) of the same trainer on torch CPU ops"
                       + (" (PointPillar front end and the frozen teacher through oracle/train_cpu.py's torch restatements, log inference off)" if what == "lidar" else "")
                       + f", batch {budget_batch}"
                       + (", 20000-point clouds" if what == "lidar" else "") + f"; {time.time() - t0:.0f} s of CPU work")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-train", action="store_true", help="skip the short train_full / train_bev runs appended to the inference line")
    ap.add_argument("--no-variants", action="store_true", help="skip the exact-fp32 child run of the frame")
    ap.add_argument("--plan-check-frames", type=int, default=40, help="frames after the timed ones whose plan is recomputed on the step-per-launch path")
    ap.add_argument("--eager", action="store_true", help="launch every kernel from Python instead of replaying HIP graphs")
    ap.add_argument("--mode", default="infer", choices=["infer", "train_full", "train_bev"],
                    help="infer: BASELINE metric (i) frames/s; train_full / train_bev: metric (ii) samples/s, data parallel")
    ap.add_argument("--batch", type=int, default=None, help="training modes: fixed per-GPU batch (weak scaling); default: BASELINE's global batch 32 / 64 split over the ranks")
    ap.add_argument("--log-every", type=int, default=100, help="train_full: steps between the log-only eval inference (1 = the reference's cadence)")
    ap.add_argument("--cpu-train-baseline", action="store_true", help="training modes: add a bounded CPU run of the same step as cpu_baseline")
    args = ap.parse_args()
    if args.mode != "infer":
        return train_bench(args)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the LAV hot path has no CPU fallback")
    if world > 1:
        # (RCCL; LAV_DIST_BACKEND=gloo lets the ranks of a test share one GPU - lav_amd/train/run.py)
        from lav_amd.train.run import setup_distributed
        _, _, device = setup_distributed()
        local = device.index
    else:
        torch.cuda.set_device(local)
        device = torch.device("cuda", local)

    from lav_amd import _lib
    lib = _lib.load()
    pipe, sds, _ = build_pipeline(device, eager=args.eager)
    host, dev = synthetic_inputs(device)
    nt = len(dev["ticks"])

    def step(i):
        loc, ori = pose(i)
        return pipe.step(dev["ticks"][i % nt], dev["all_rgbs"], dev["rgbs"], dev["tel_rgbs"], loc, ori, dev["nxp"], 3)

    if not args.eager:
        pipe.precapture(cmds=[3], max_others=8)   # no HIP-graph capture inside the timed region, whatever the detections do
    # fill the 15-frame history + warm up (untimed); profiling is armed here so its event pools are created now
    if args.eager:
        lib.lav_profile_enable(min(65000, 200 * args.steps))
    i = 0
    # the agent stacks the sweeps of frames t, t-5 and t-10: the first frame whose three sweeps all exist is the 16th,
    # so fewer than 16 untimed frames would time a lighter workload.  The line reports the number actually run.
    n_warm = max(args.warmup, 16)
    for _ in range(n_warm):
        step(i); i += 1
    torch.cuda.synchronize()

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()

    lib.lav_profile_reset()
    health0 = pipe.health() if hasattr(pipe, "health") else None
    barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step(i); i += 1
    torch.cuda.synchronize(); barrier()
    dt = time.perf_counter() - t0
    # every timed frame's outputs were finite and no persistent plan launch gave up: read from the pipeline's sticky device-side
    # counters AFTER the timed region (the graphs carry the checks; nothing was copied up per frame)
    health = None
    if health0 is not None:
        h1 = pipe.health()
        health = {k: h1[k] - health0[k] for k in ("nonfinite_outputs", "finite_checks", "plan_launches", "plan_aborts", "plans_recomputed",
                                                  "decode_mismatches", "overflow_ticks", "pair_chain_launches", "pair_chain_timeouts")}
        health["last_plan_launch"] = h1["last_plan_launch"]
    host_finite = all(bool(torch.isfinite(out[k]).all()) for k in ("ego_plan_locs", "ego_cast_locs", "ego_embd", "pred_bra", "pred_bev")) \
        and bool(torch.isfinite(out["other_cast_locs"]).all())
    # finite is not yet right: more frames (untimed, --plan-check-frames), each plan recomputed on the step-per-launch path and compared
    plan_dev = None
    if hasattr(pipe, "plan_deviation"):
        plan_dev = 0.0
        for _ in range(args.plan_check_frames):
            o = step(i); i += 1
            plan_dev = max(plan_dev, pipe.plan_deviation(o, 3))
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    def read(name):
        ms, n = ctypes.c_double(), ctypes.c_int()
        lib.lav_profile_read(name.encode(), ctypes.byref(ms), ctypes.byref(n))
        return ms.value, n.value
    prof = {k: read(k) for k in ("pointnet_scatter", "pillar_prep", "conv2d", "paint", "gru_cast", "gru_plan")}
    lib.lav_profile_enable(0)

    n_pts = int(out["lidar_points"].shape[0])
    per_frame_us = {k: round(v[0] / args.steps * 1e3, 1) for k, v in prof.items()} if args.eager else None

    def pillar_micro(n_per_sweep, reps=100, batch=1):
        """BASELINE.json config #2 in isolation: back-to-back pillar launches on one synthetic cloud, so the
        HIP-event figures are pure kernel time (the frame loop above is host-bound between launches).
        batch > 1 (round 6): that many clouds in ONE call - what the kernel pair costs per canvas once its launch chain is amortised."""
        from lav_amd import synth
        ppn1 = pipe.infer_model.lidar_model.point_pillar_net
        if batch > 1:
            clouds = [torch.from_numpy(synth.stacked_lidar(n_per_sweep, seed=synth.SEED + b)).to(device) for b in range(batch)]
            ppn = lambda lst, ns: ppn1(clouds, [len(c) for c in clouds])
            pts = torch.cat(clouds)
        else:
            pts = torch.from_numpy(synth.stacked_lidar(n_per_sweep)).to(device)
            ppn = ppn1
        for _ in range(5):
            ppn([pts], [len(pts)])
        torch.cuda.synchronize()
        lib.lav_profile_enable(reps + 8)
        ppn([pts], [len(pts)])
        torch.cuda.synchronize()
        lib.lav_profile_reset()
        for _ in range(reps):
            ppn([pts], [len(pts)])
        torch.cuda.synchronize()
        k_ms, k_n = read("pointnet_scatter")
        p_ms, p_n = read("pillar_prep")
        lib.lav_profile_enable(0)
        # the stage as the stream sees it: ONE event pair around `reps` back-to-back (k_bin, k_rows) pairs - no per-kernel event
        # records between the launches (each costs the kernel it brackets ~2-3 us of the figures above)
        # (replayed from a HIP graph of 20 pairs: launched from Python the region would measure the host)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            ppn([pts], [len(pts)])
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        in_graph = 20 if batch == 1 else 2   # (every call inside the capture allocates its canvases in the graph's pool: 0.4 GB each at 16 clouds)
        with torch.cuda.graph(gr, stream=side):
            for _ in range(in_graph):
                ppn([pts], [len(pts)])
        gr.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            gr.replay()
        e1.record(); torch.cuda.synchronize()
        region_us = e0.elapsed_time(e1) / (5 * in_graph) * 1e3
        del gr
        nb = 4 * (len(pts) * 11 + batch * 64 * 320 * 320)
        ks, ps = k_ms / max(k_n, 1) * 1e-3, p_ms / max(p_n, 1) * 1e-3
        kept = int(((pts[:, 0] >= -10) & (pts[:, 0] < 70) & (pts[:, 1] >= -40) & (pts[:, 1] < 40)).sum())
        fl = 10240.0 * kept    # PointNet: 2*(16*64 + 64*64) flop per kept point, on the matrix cores inside the same kernel
        return dict(points=len(pts), algorithmic_bytes=nb, kernel_us=round(ks * 1e6, 2), prep_us=round(ps * 1e6, 2),
                    achieved=round(nb / ks / 1e9, 1), frac=round(nb / ks / 1e9 / HBM_PEAK_GBS, 4),
                    pipeline_achieved=round(nb / (ks + ps) / 1e9, 1), unit="GB/s", peak=HBM_PEAK_GBS,
                    stage_region_us=round(region_us, 2), stage_region_frac=round(nb / (region_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                    pointnet_flops=fl, pointnet_tflops=round(fl / ks / 1e12, 1),
                    mfma_bound_us=round(fl / (MFMA_F32_PEAK_TFLOPS * 1e12) * 1e6, 2), hbm_bound_us=round(nb / (HBM_PEAK_GBS * 1e9) * 1e6, 2))
    def conv_micro(reps=50):
        """The frame's largest dense kernel in isolation: the fused 384->256 3x3 convolution of the four heads on the
        (1,384,160,160) feature map - MFMA-bound, fp32 matrix peak 157.3 TFLOP/s."""
        from lav_amd import ops
        lm = pipe.infer_model.lidar_model
        feats = torch.randn((1, 384, 160, 160), device=device)
        with ops.precision(pipe.precision):     # the engine the frame's heads graph runs (round 6: LAV_CONV_F16X3 through ops.precision)
            lm.heads(feats)
            layer = lm._head_engine(lm.ALL_HEADS, device)["conv"]
        # the scale of the activations as the frame hands it over: maxima left by the launches that wrote the feature map (here: one part
        # holding the tensor's maximum) - round 5 timed the measuring launch with the convolution, the frame no longer has one
        am = ops.Amax(device)
        am.take(4)[0] = feats.abs().max()
        for _ in range(3):
            layer(feats, amax_in=am)
        torch.cuda.synchronize()
        lib.lav_profile_enable(reps + 8)
        layer(feats, amax_in=am); torch.cuda.synchronize(); lib.lav_profile_reset()
        for _ in range(reps):
            layer(feats, amax_in=am)
        torch.cuda.synchronize()
        ms, n = read("conv2d")
        lib.lav_profile_enable(0)
        flops = 2.0 * 160 * 160 * 256 * 384 * 9
        sec = ms / max(n, 1) * 1e-3
        desc = _lib.Conv.from_buffer_copy(layer.desc); desc.batch, desc.h, desc.w = 1, 160, 160
        info = (ctypes.c_int * 9)()
        lib.lav_conv_tile_info(ctypes.byref(desc), info)
        # Calibration of the roof (round 5): the chip is POWER bound under dense bf16 matrix work (tools/clock_probe.py: ~1.33-1.39 kW of
        # the 1.4 kW socket limit, shader clock 1.9-2.1 GHz instead of 2.4), so even the vendor's plain bf16 GEMM stays well below the
        # 2.5 PFLOP/s of the guide.  hipBLASLt's 8192^3 bf16 matmul is timed here, in the same process, as the rate a matrix kernel can
        # sustain on this box; `frac` stays against the guide's peak (the contract), `frac_of_vendor_gemm` is the second reading.
        vend = None
        try:
            am = torch.randn((8192, 8192), device=device, dtype=torch.bfloat16)
            for _ in range(3):
                am @ am
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                am @ am
            e1.record(); torch.cuda.synchronize()
            vend = 2.0 * 8192 ** 3 / (e0.elapsed_time(e1) / 10 * 1e-3) / 1e12
            del am
        except Exception:  # noqa: BLE001 - a calibration line, never a reason to lose the bench
            vend = None
        if info[0] == -1:   # split kernel: six bf16 MFMA products per fp32 product - or, LAV_CONV_F16X3 (round 5: what the frame pipelines
            # ask for on this layer), three fp16 ones, with the launch that measures the activation scale inside the timed pair
            f16 = info[7] >= 200
            executed = (3.0 if f16 else 6.0) * flops
            return dict(bound="mfma", kernel=("k_conv_split_f16<2,2> heads 384->256 3x3 @160x160 (f16x3: two fp16 pieces per operand, three products; activation scale from the producers' maxima)"
                                              if f16 else "k_conv_split<2,2> heads 384->256 3x3 @160x160 (bf16x6 split operands)"),
                        executed_per_algorithmic=3 if f16 else 6,
                        vendor_gemm_tflops=None if vend is None else round(vend, 1),
                        vendor_gemm="torch bf16 8192^3 matmul (hipBLASLt), 10 launches after 3 warm-ups, same process: the sustained bf16 rate of this box (power bound)",
                        frac_of_vendor_gemm=None if vend is None else round(executed / sec / 1e12 / vend, 4),
                        achieved=round(executed / sec / 1e12, 1), peak=MFMA_BF16_PEAK_TFLOPS, unit="TFLOP/s (16-bit MFMA flops executed: executed_per_algorithmic x 2MNK)",
                        frac=round(executed / sec / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4), traffic=None, algorithmic_flops=flops,
                        fp32_equivalent_tflops=round(flops / sec / 1e12, 1),
                        vs_fp32_mfma_peak=round(flops / sec / 1e12 / MFMA_F32_PEAK_TFLOPS, 3), avg_kernel_us=round(sec * 1e6, 1), launches=reps)
        return dict(bound="mfma", kernel="k_conv<2,2> heads 384->256 3x3 @160x160 (fp32 MFMA)", achieved=round(flops / sec / 1e12, 1),
                    peak=MFMA_F32_PEAK_TFLOPS, unit="TFLOP/s", frac=round(flops / sec / 1e12 / MFMA_F32_PEAK_TFLOPS, 4),
                    traffic=None, algorithmic_flops=flops, avg_kernel_us=round(sec * 1e6, 1), launches=reps)

    def glue_micro(reps=50):
        """The other HBM-bound hand kernels SURVEY 8(d) names, in isolation (back-to-back launches, library HIP events):
        painting, temporal stacking + history write, rotated crop.  Algorithmic bytes as SURVEY 8(d) counts them."""
        from lav_amd import ops, synth
        outm = {}

        def timed(name, fn):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            lib.lav_profile_enable(reps + 8)
            fn(); torch.cuda.synchronize(); lib.lav_profile_reset()
            for _ in range(reps):
                fn()
            torch.cuda.synchronize()
            ms, n = read(name)
            lib.lav_profile_enable(0)
            return ms / max(n, 1) * 1e-3

        def entry(kernel, nbytes, sec, what):
            return dict(bound="hbm", kernel=kernel, achieved=round(nbytes / sec / 1e9, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                        frac=round(nbytes / sec / 1e9 / HBM_PEAK_GBS, 4), algorithmic_bytes=int(nbytes), avg_kernel_us=round(sec * 1e6, 2),
                        launches=reps, bytes_counted=what, traffic=None)
        im = pipe.infer_model
        n_pt = 65536
        cur = torch.from_numpy(np.concatenate([host["ticks"][0], host["ticks"][1]])).to(device)
        sem = torch.softmax(torch.randn((3, 5, 288, 256), device=device), dim=1)
        sec = timed("paint", lambda: im.forward_paint(cur, sem))
        outm["paint"] = entry("k_paint (3 cameras, class-0 suppression, concat)", n_pt * (16 + 32) + sem.numel() * 4, sec,
                              "N x (16 B point read + 32 B fused row written) + the 3x5x288x256 probability maps, N = 65536")
        fused = torch.randn((n_pt, 8), device=device)
        ring = torch.randn((15, n_pt, 8), device=device)
        slot = torch.tensor([3], device=device); sweeps = torch.tensor([3, 13, 8], device=device)
        R = torch.eye(3, device=device)[None].repeat(3, 1, 1).contiguous(); t = torch.zeros((3, 3), device=device)
        sec = timed("stack_sweeps", lambda: ops.stack_sweeps(fused, ring, slot, sweeps, R, t))
        outm["stack_sweeps"] = entry("k_stack_sweeps (history write + 3-sweep stack in the ego frame)", n_pt * 32 * 2 + 2 * n_pt * 32 + 3 * n_pt * 44,
                                     sec, "new sweep read + written to the ring (2 x 32 B), two old sweeps read (32 B), 3 x N rows of 44 B written, N = 65536")
        feats = torch.randn((1, 384, 160, 160), device=device)
        for n_c in (1, 4):
            locs = torch.tensor([[3.0 * i, -5.0 - 2 * i] for i in range(n_c)], device=device)
            oris = torch.tensor([0.3 * i for i in range(n_c)], device=device)
            sec = timed("crop_rotate", lambda: ops.crop_rotate(feats, locs, oris, 2.0, 96, 0.0, 0.75))
            outm[f"crop_rotate_n{n_c}"] = entry("k_crop_rotate_staged (affine_grid + bilinear grid_sample, 384 channels)", n_c * 384 * 96 * 96 * 4 * 2, sec,
                                              f"{n_c} crop(s): every output value written once + the map region under the crop read once (crop pitch ~ map pitch)")
        return outm

    micro = {"config2_32768pts": pillar_micro(10923), "agent_196608pts": pillar_micro(65536)} if rank == 0 else None
    if micro is not None:
        try:   # sixteen config-#2 clouds in one call: the same kernels with the launch chain paid once
            micro["config2_batch16_one_call"] = dict(pillar_micro(10923, reps=30, batch=16), clouds=16,
                                                     note="16 clouds of 32 769 points, one lav_pillar_scatter call: the kernel pair's throughput regime - not BASELINE config #2, which is ONE cloud (roofline.frac)")
        except Exception as e:   # never lose the headline line to a side measurement
            micro["config2_batch16_one_call"] = dict(error=repr(e)[:200])
    conv_roof = conv_micro() if rank == 0 else None
    # roofline of the dominant pillar kernel at the frame's own size (196 608 points).  The frame loop replays HIP
    # graphs (kernels inside a graph cannot carry event pairs), so the figure comes from the back-to-back launches
    # above on the same library stream: pure kernel time, no launch gaps.
    roofline = None
    glue = glue_micro() if rank == 0 else None
    if rank == 0:
        # The north-star figure: fraction of the HBM roof (algorithmic bytes 4 (N D + C ny nx)) of the pillar kernel,
        # at BASELINE config #2 (32 768-point sweep); the frame's own 196 608-point stack and the matrix-pipe term of the fused
        # PointNet (max(bytes / 8 TB/s, flops / 157.3 TFLOP/s) is the binding roof) are carried beside it.
        m, mf = micro["config2_32768pts"], micro["agent_196608pts"]
        traffic, traffic_src = None, None
        for prof_name in ("r06_pmc_pillar.json", "r05_pmc_pillar.json", "r04_pmc_pillar.json", "r03_pmc_pillar.json", "r02_b_pmc_pillar.json"):
            try:  # HBM bytes per launch from the committed PMC pass of this kernel (counters cannot be read from inside a run)
                with open(os.path.join(REPO, "profiles", prof_name)) as f:
                    pm = json.load(f)
                traffic = pm["k_rows"][str(m["points"])]["traffic_bytes"]
                traffic_src = f"profiles/{prof_name} (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate passes)"
                break
            except Exception:
                continue
        roofline = dict(bound="hbm", kernel="k_rows (pillar PointNet + scatter-max + canvas), BASELINE config #2", achieved=m["achieved"], peak=HBM_PEAK_GBS,
                        unit="GB/s", frac=m["frac"], algorithmic_bytes=m["algorithmic_bytes"], traffic=traffic, traffic_source=traffic_src,
                        points=m["points"], avg_kernel_us=m["kernel_us"], launches=100, hbm_bound_us=m["hbm_bound_us"],
                        stage_us=round(m["kernel_us"] + m["prep_us"], 2), stage_frac=round(m["pipeline_achieved"] / HBM_PEAK_GBS, 4),
                        stage_region_us=m["stage_region_us"], stage_region_frac=m["stage_region_frac"],
                        frac_mfma=round(m["pointnet_tflops"] / MFMA_F32_PEAK_TFLOPS, 4), mfma_bound_us=m["mfma_bound_us"],
                        frame_cloud=dict(points=mf["points"], avg_kernel_us=mf["kernel_us"], frac_hbm=mf["frac"], achieved_hbm_GBs=mf["achieved"],
                                         frac_mfma=round(mf["pointnet_tflops"] / MFMA_F32_PEAK_TFLOPS, 4), hbm_bound_us=mf["hbm_bound_us"],
                                         mfma_bound_us=mf["mfma_bound_us"], stage_us=round(mf["kernel_us"] + mf["prep_us"], 2)),
                        bound_rule="max(algorithmic bytes / 8 TB/s, algorithmic PointNet flops / 157.3 TFLOP/s), SURVEY 8(d); frac is the bytes term")

    def forced_frames(n_forced, steps=40):
        """Frame time with the others branch forced to n fixed poses."""
        nonlocal i
        locs = [[4.0 + 3.0 * k, -8.0 - 4.0 * k] for k in range(n_forced)]
        oris = [0.2 * k - 0.3 for k in range(n_forced)]
        pipe.set_forced_others(locs, oris)
        for _ in range(6):
            step(i); i += 1
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step(i); i += 1
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        return dict(ms_per_step=round(ms, 4), frames_per_s=round(1e3 / ms, 2), steps=steps)
    def upload_frames(steps=60):
        """the reference's run_step uploads the tick's sensor data (lav_agent_fast.py:233,263,311-312); the
        headline keeps the inputs resident in HBM (the contract's definition of `value`).  Here the same frames are fed from PINNED
        HOST tensors, so that the five host-to-device copies are inside the step; plus the copies alone, and the number of copy
        launches a frame issues either way (counted at Tensor.copy_)."""
        nonlocal i
        pin = dict(ticks=[torch.from_numpy(t).pin_memory() for t in host["ticks"]], all_rgbs=torch.from_numpy(host["all_rgbs"]).pin_memory(),
                   rgbs=torch.from_numpy(host["rgbs"]).pin_memory(), tel_rgbs=torch.from_numpy(host["tel_rgbs"]).pin_memory(),
                   nxp=torch.from_numpy(host["nxp"]).pin_memory())
        nbytes = sum(t.numel() * t.element_size() for t in (pin["ticks"][0], pin["all_rgbs"], pin["rgbs"], pin["tel_rgbs"], pin["nxp"]))

        def step_h(j):
            loc, ori = pose(j)
            return pipe.step(pin["ticks"][j % nt], pin["all_rgbs"], pin["rgbs"], pin["tel_rgbs"], loc, ori, pin["nxp"], 3)
        for _ in range(6):
            step_h(i); i += 1
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step_h(i); i += 1
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        # the copies alone, on an idle GPU
        bufs = [torch.empty_like(t, device=device) for t in (pin["ticks"][0], pin["all_rgbs"], pin["rgbs"], pin["tel_rgbs"], pin["nxp"])]
        srcs = [pin["ticks"][0], pin["all_rgbs"], pin["rgbs"], pin["tel_rgbs"], pin["nxp"]]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(50):
            for b, t in zip(bufs, srcs):
                b.copy_(t, non_blocking=True)
        e1.record(); torch.cuda.synchronize()
        h2d_ms = e0.elapsed_time(e1) / 50
        # copy launches per frame, counted where they are issued
        counts = {}
        orig = torch.Tensor.copy_

        def counting(self, src, *a, **k):
            kind = ("h" if not src.is_cuda else "d") + "2" + ("h" if not self.is_cuda else "d")
            counts[kind] = counts.get(kind, 0) + 1
            return orig(self, src, *a, **k)
        per_frame = {}
        for label, fn in (("resident_inputs", step), ("host_inputs", step_h)):
            counts.clear()
            torch.Tensor.copy_ = counting
            try:
                for _ in range(10):
                    fn(i); i += 1
            finally:
                torch.Tensor.copy_ = orig
            torch.cuda.synchronize()
            per_frame[label] = {k: round(counts.get(k, 0) / 10, 1) for k in ("d2h", "h2d")}   # (round 6: 0 / 0 with resident inputs - the peak rows
            #  reach the host through lav_det_decode_report's stores into pinned memory, the pose block rides in the staging launch's arguments)
        return dict(ms_per_step=round(ms, 4), frames_per_s=round(1e3 / ms, 2), steps=steps, bytes_per_frame=int(nbytes),
                    h2d_ms_per_frame=round(h2d_ms, 4), h2d_GBs=round(nbytes / h2d_ms / 1e6, 1),
                    note="float32 camera tensors (7.0 MB per frame: an upper bound - the agent itself uploads uint8 images, 3.2 MB); pinned host memory",
                    copy_launches_per_frame=per_frame)
    uploads = None
    if rank == 0 and world == 1 and not args.eager:
        try:
            uploads = upload_frames()
        except Exception as e:   # never lose the headline line to a side measurement
            uploads = dict(error=repr(e)[:200])
    forced = None
    if rank == 0 and world == 1 and not args.eager:
        forced = {f"others_{k}": forced_frames(k) for k in (0, 4)}
        pipe.set_forced_others(None)

    def chain_only(steps=40):
        """The frame with the two side streams (brake net, ego branch) switched off: the critical chain lidar -> heads -> others alone
        (their outputs keep their last values).  frame - chain = what the side streams' contention costs."""
        nonlocal i
        from lav_amd import frame as frame_mod
        saved = frame_mod._DIAG_SKIP
        frame_mod._DIAG_SKIP = {"brake", "ego"}
        try:
            for _ in range(6):
                step(i); i += 1
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                step(i); i += 1
            torch.cuda.synchronize()
            return round((time.perf_counter() - t0) / steps * 1e3, 4)
        finally:
            frame_mod._DIAG_SKIP = saved

    def graph_times(iters=50):
        """Stand-alone replay time of each frame graph (nothing else on the GPU): what the streams would take one after the other."""
        res = {}
        for key, g in pipe.graphs.items():
            name = key if isinstance(key, str) else "_".join(str(k) for k in key)
            state = (pipe.ring.clone(), pipe.b_prev.clone())
            torch.cuda.synchronize()
            for _ in range(3):
                g.replay()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(iters):
                g.replay()
            torch.cuda.synchronize()
            res[name] = round((time.perf_counter() - t0) / iters * 1e3, 4)
            pipe.ring.copy_(state[0]); pipe.b_prev.copy_(state[1])
        return res
    chain_ms = graphs_ms = None
    if rank == 0 and world == 1 and not args.eager:
        chain_ms = chain_only()
        graphs_ms = graph_times()

    def f32_frame():
        """The same frame with every convolution on the exact-fp32 MFMA kernels (LAV_CONV_PRECISION=f32 is read once per process):
        a child run of this script, inference line only."""
        import subprocess
        env = dict(os.environ, LAV_CONV_PRECISION="f32")
        cmd = [sys.executable, os.path.abspath(__file__), "--steps", "40", "--warmup", "16", "--no-train", "--no-cpu-baseline", "--no-variants"]
        try:
            r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
            line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
            return dict(ms_per_step=line["ms_per_step"], frames_per_s=line["value"], steps=line["steps"], health=line.get("health"),
                        precision="LAV_CONV_PRECISION=f32: v_mfma_f32_32x32x2_f32 everywhere (bit-for-bit fmaf chains)")
        except Exception as e:   # never lose the headline line to the variant
            return dict(error=repr(e)[:200])

    training = None
    if world > 1 and not args.no_train:
        # the data-parallel training lines of this N: every rank runs a child; this process's group and frame memory go first
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
        torch.cuda.empty_cache()
        training = training_lines(with_cpu=False, world=world, rank=rank)
    if rank == 0:
        res = dict(metric="frames/s full agent fwd (32k-pt LiDAR + 3 cams)", value=round(world * args.steps / dt, 2),
                   unit="frames/s", n_gpus=world, steps=args.steps, warmup=n_warm,
                   ms_per_step=round(dt / args.steps * 1e3, 4), higher_is_better=True, scaling="weak", vs_baseline=None,
                   dtype=DTYPE, data="synthetic",
                   config=dict(workload="full lav_agent_fast forward, batch 1: 2x32768-pt half sweeps -> 3-sweep stack "
                                        f"({n_pts} pts x 11) + 3x288x256 RGB + 288x480 tele; ERFNet seg, paint, pillar 320x320x64, "
                                        "BEV backbone+heads, uniplanner (cast+plan GRUs), brake net",
                               parallelism=f"replicas x{world}" if world > 1 else "single GPU", vehicles_detected=len(out["det"][1]),
                               launch="eager" if args.eager else "hip graphs: lidar / heads / others (capacity 15, device-resident count) on the main stream, brake_a (with the frame) + brake_b (behind the lidar graph) and ego[cmd] on side streams"),
                   roofline=roofline, roofline_pillar_isolated=micro, roofline_mfma=conv_roof, roofline_hbm_glue=glue,
                   forced_others=forced, hip_kernel_us_per_frame=per_frame_us,
                   health=health, last_frame_outputs_finite=host_finite, plan_vs_step_path_max_abs=plan_dev, chain_only_ms=chain_ms,
                   graph_replay_ms=graphs_ms, with_sensor_upload=uploads)
        bad = [] if host_finite else ["the last timed frame holds non-finite outputs"]
        if health is not None:
            if health["nonfinite_outputs"]:
                bad.append(f"{health['nonfinite_outputs']} non-finite output tensors inside the timed frames")
            if health["plan_aborts"] or health["plans_recomputed"]:
                bad.append(f"{health['plan_aborts']} persistent plan launches timed out")
            if health["pair_chain_timeouts"]:
                bad.append(f"{health['pair_chain_timeouts']} workgroups of the persistent ERFNet runs gave up waiting")
        if plan_dev is not None and not plan_dev <= 1e-5:
            bad.append(f"the graphs' plan differs from the step path by {plan_dev:.3g} m (finite but wrong)")
        res["valid"] = not bad
        if bad:
            res["invalid_reason"] = "; ".join(bad)
        if world == 1 and not args.eager and not args.no_variants:
            res["frame_fp32_kernels"] = f32_frame()
        if world == 1 and not args.no_train:
            res["training"] = training_lines(with_cpu=not args.no_cpu_baseline)
        elif training is not None:
            res["training"] = training
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(sds, host)
        else:
            res["cpu_baseline"] = None
        print(json.dumps(res))
    if world > 1:
        import torch.distributed as dist
        if dist.is_initialized():
            dist.destroy_process_group()


if __name__ == "__main__":
    main()
