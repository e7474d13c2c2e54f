#!/usr/bin/env python3
"""Per-layer time of every lav_conv2d call of one network (eager, events around each call, best of several runs).

    python tools/layer_times.py seg|brake|ego|bev
"""
import ctypes
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402
from lav_amd import _lib, ops  # noqa: E402
from lav_amd._lib import Conv  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "seg"
dev = torch.device("cuda:0")
pipe, sds, (lm, up, seg, bra) = bench.build_pipeline(dev, eager=True)
lib = _lib.load()
records = []
orig = ops.ConvLayer.__call__


def hooked(self, x, out=None, residual=None, **kw):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    y = orig(self, x, out=out, residual=residual, **kw)
    e1.record()
    records.append((self, tuple(x.shape), e0, e1))
    return y


ops.ConvLayer.__call__ = hooked
pair_orig = ops.Conv1dPair.__call__


def pair_hooked(self, x, residual=None):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    y = pair_orig(self, x, residual=residual)
    e1.record()
    pairs.append((self, tuple(x.shape), e0, e1))
    return y


pairs = []
ops.Conv1dPair.__call__ = pair_hooked
with torch.no_grad():
    if which == "seg":
        x = torch.rand((3, 3, 288, 256), device=dev) * 255
        fn = lambda: seg(x)
    elif which == "brake":
        x1, x2 = torch.rand((1, 3, 288, 768), device=dev) * 255, torch.rand((1, 3, 192, 480), device=dev) * 255
        fn = lambda: bra(x1, x2)
    elif which == "ego":
        x = torch.randn((int(sys.argv[2]) if len(sys.argv) > 2 else 1, 384, 96, 96), device=dev)
        fn = lambda: up.lidar_conv_emb(x)
    else:
        x = torch.randn((1, 64, 320, 320), device=dev)
        fn = lambda: lm.heads(lm.backbone(x))
    best = None
    pbest = None
    for it in range(6):
        records.clear(); pairs.clear()
        with ops.precision(pipe.precision):     # (the engines the frame runs: LAV_CONV_F16X3 unless LAV_INFER_PRECISION says otherwise)
            fn()
        torch.cuda.synchronize()
        t = [r[2].elapsed_time(r[3]) * 1e3 for r in records]
        best = t if best is None else [min(a, b) for a, b in zip(best, t)]
        pt = [r[2].elapsed_time(r[3]) * 1e3 for r in pairs]
        pbest = pt if pbest is None else [min(a, b) for a, b in zip(pbest, pt)]
    tot = 0
    for (layer, shp, _, _), us in zip(records, best):
        d = Conv.from_buffer_copy(layer.desc); d.batch, d.h, d.w = shp[0], shp[2], shp[3]
        info = (ctypes.c_int * 9)()
        lib.lav_conv_tile_info(ctypes.byref(d), info)
        oh, ow = layer.out_hw(shp[2], shp[3])
        flops = 2.0 * shp[0] * oh * ow * d.cout * d.cin * d.kh * d.kw / (d.stride ** 2 if d.transposed else 1)
        nch = (d.cin + 15) // 16
        print(f"{'T' if d.transposed else 'C'} {d.cin:4d}->{d.cout:4d} k{d.kh}x{d.kw} s{d.stride} d{d.dil_h},{d.dil_w} in {shp[0]}x{shp[2]}x{shp[3]:4d} "
              f"{us:7.1f} us {flops / us / 1e6:6.1f} TF/s  {info[0]}x{info[1]} rb={info[2]} lds={info[5] // 1024:3d}K ks={info[6]} tg={info[7]} cps={info[8]} "
              f"stages={-(-(-(-nch // max(info[6], 1))) // max(info[8], 1))}")
        tot += us
    print(f"total {tot:.0f} us over {len(records)} conv launches")
    if pairs:
        for (layer, shp, _, _), us in zip(pairs, pbest):
            print(f"P {layer.ch:4d} d{layer.da:2d} in {shp[0]}x{shp[2]}x{shp[3]:4d} {us:7.1f} us")
        print(f"pairs total {sum(pbest):.0f} us over {len(pairs)} launches")
