#!/usr/bin/env python3
"""Stand-alone replay time of each chain of the frame graph (ERFNet, lidar chain, heads+peaks, ego branch, brake
trunks, others branch) - what an ideal overlap could reach.  GPU only."""
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402
from lav_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
pipe, sds, (lm, up, seg, bra) = bench.build_pipeline(dev)
host, d = bench.synthetic_inputs(dev)
for i in range(25):
    loc, ori = bench.pose(i)
    pipe.step(d["ticks"][i % 4], d["all_rgbs"], d["rgbs"], d["tel_rgbs"], loc, ori, d["nxp"], 3)
torch.cuda.synchronize()


def timeit(name, fn, iters=100):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        g.replay()
    torch.cuda.synchronize()
    print(f"{name:28s} {(time.perf_counter() - t0) / iters * 1e3:8.3f} ms", flush=True)


with torch.no_grad():
    im = pipe.infer_model
    cur = ops.merge_ticks(pipe.b_tick, pipe.b_prev.clone())
    pred_sem = torch.softmax(seg(pipe.b_all_rgbs), dim=1)
    fused = im.forward_paint(cur, pred_sem)
    pts = ops.stack_sweeps(fused, pipe.ring.clone(), pipe.b_slot, pipe.b_sweeps, pipe.b_R, pipe.b_t)
    canvas = lm.point_pillar_net([pts], [pts.shape[0]])
    feats = lm.backbone(canvas)
    timeit("erfnet+softmax", lambda: torch.softmax(seg(pipe.b_all_rgbs), dim=1))
    timeit("paint", lambda: im.forward_paint(cur, pred_sem))
    timeit("pillar", lambda: lm.point_pillar_net([pts], [pts.shape[0]]))
    timeit("bev backbone", lambda: lm.backbone(canvas))
    timeit("heads", lambda: lm.heads(feats))
    heat, size, ori, bev = lm.heads(feats)
    timeit("peaks", lambda: ops.extract_peaks(heat[0], size[0], ori[0], apply_sigmoid=True))
    z2, z1 = torch.zeros((1, 2), device=dev), torch.zeros((1,), device=dev)
    timeit("ego crop", lambda: up.crop_feature(feats, z2, z1, up.pixels_per_meter / 2, up.crop_size))
    crop = up.crop_feature(feats, z2, z1, up.pixels_per_meter / 2, up.crop_size)
    timeit("ego resnet18", lambda: up.lidar_conv_emb(crop))
    embd = up.lidar_conv_emb(crop)
    timeit("ego cast", lambda: up.cast(embd, mode="ego"))
    cast = up.cast(embd, mode="ego")
    timeit("ego plan", lambda: up.plan(embd, pipe.b_nxp[None], cast_locs=cast, pixels_per_meter=up.pixels_per_meter,
                                       crop_size=up.crop_size * 2, cmd=3))
    timeit("brake wide trunk", lambda: bra.conv_backbone(bra.normalize(pipe.b_rgbs / 255.)))
    timeit("brake tele trunk", lambda: bra.conv_backbone(bra.normalize(pipe.b_tel / 255.)))
    x1 = bra.conv_backbone(bra.normalize(pipe.b_rgbs / 255.)); x2 = bra.conv_backbone(bra.normalize(pipe.b_tel / 255.))
    timeit("brake attn+classifier", lambda: bra.classifier(torch.cat([bra.attn1(x1), bra.attn2(x2)], dim=1)))
    for n in (1, 2, 4):
        pipe.b_locs[:n] = torch.tensor([[5.0, 1.0]] * n, device=dev)
        timeit(f"others graph N={n}", lambda: pipe._g_others(n))
    timeit("lidar graph", pipe._g_lidar)
    timeit("heads graph", pipe._g_heads)
    timeit("ego graph", lambda: pipe._g_ego(3))
    timeit("brake graph", pipe._g_brake)
