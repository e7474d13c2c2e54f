#!/usr/bin/env python3
"""Every convolution / BatchNorm shape one train_full_v2 step runs (forward hooks on the trained models), and for each
unique convolution: torch (MIOpen) forward / data gradient / weight gradient against lav_conv2d forward and its transposed
plan as the data gradient.

    python tools/train_conv_probe.py [batch]
"""
import os
import sys
import time

import torch
import torch.nn as nn
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lav_amd.ops import ConvLayer  # noqa: E402
from lav_amd.train import LAV, TrainConfig, synthetic_lidar_batch  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
SKIP = int(sys.argv[2]) if len(sys.argv) > 2 else 0          # skip the first SKIP shapes (to get at a late one quickly)
VERBOSE = len(sys.argv) > 3                                   # print every sub-step before it runs (locating a fault)
say = (lambda *a: print("   ...", *a, flush=True)) if VERBOSE else (lambda *a: None)
dev = torch.device("cuda")
lav = LAV(TrainConfig(), dev, what="lidar")
batch = synthetic_lidar_batch(B, device=dev)
seen = {}


def hook(m, inp, out):
    x = inp[0]
    if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
        key = (type(m).__name__, m.in_channels, m.out_channels, m.kernel_size, m.stride, m.padding, m.dilation,
               getattr(m, "output_padding", (0, 0)), m.bias is not None, tuple(x.shape))
    else:
        key = ("BatchNorm2d", tuple(x.shape))
    seen[key] = seen.get(key, 0) + (1 if torch.is_grad_enabled() and x.requires_grad or isinstance(m, nn.BatchNorm2d) and m.training else 0)


hs = []
for root in (lav.lidar_model, lav.uniplanner):
    for m in root.modules():
        if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d, nn.BatchNorm2d)):
            hs.append(m.register_forward_hook(hook))
lav.train_lidar(*batch)
torch.cuda.synchronize()
for h in hs:
    h.remove()


def ev(fn, reps=3):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


tot = dict(fwd=0.0, dgrad=0.0, wgrad=0.0, lav_fwd=0.0, lav_dgrad=0.0, bn=0.0)
print(f"batch {B}: unique shapes {len(seen)}", flush=True)
for idx, (key, cnt) in enumerate(sorted(seen.items(), key=lambda kv: -kv[1])):
    if cnt == 0 or idx < SKIP:
        continue
    say(idx, key)
    if key[0] == "BatchNorm2d":
        x = torch.randn(key[1], device=dev)
        bn = nn.BatchNorm2d(key[1][1]).to(dev).train()
        xg = x.clone().requires_grad_(True)
        def step():
            y = F.relu(bn(xg))
            y.backward(x)
        t = ev(step)
        tot["bn"] += t * cnt
        print(f"x{cnt:3d} BN+ReLU fwd+bwd {str(key[1]):28s} {t:9.1f} us  ({x.numel() * 4 * 8 / t / 1e6:5.2f} TB/s at 8 passes)")
        continue
    name, cin, cout, k, s, p, d, op, has_bias, xs = key
    tr = name == "ConvTranspose2d"
    x = torch.randn(xs, device=dev)
    w = torch.randn((cin, cout, *k) if tr else (cout, cin, *k), device=dev) / (cin * k[0] * k[1]) ** 0.5
    fn = (lambda a, ww: F.conv_transpose2d(a, ww, None, s, p, op, 1, d)) if tr else (lambda a, ww: F.conv2d(a, ww, None, s, p, d))
    say("torch forward")
    y = fn(x, w)
    torch.cuda.synchronize()
    dy = torch.randn_like(y)
    flops = 2.0 * y.numel() * cin * k[0] * k[1] / ((s[0] * s[1]) if tr else 1)
    t_f = ev(lambda: fn(x, w))
    xg = x.clone().requires_grad_(True)
    wg = w.clone().requires_grad_(True)
    yx = fn(xg, w)
    say("torch dgrad")
    t_d = ev(lambda: torch.autograd.grad(yx, xg, dy, retain_graph=True))
    yw = fn(x, wg)
    say("torch wgrad")
    t_w = ev(lambda: torch.autograd.grad(yw, wg, dy, retain_graph=True))
    # lav forward, and the data gradient as the adjoint plan
    lf = ConvLayer(w.cpu(), stride=s[0], padding=p, dilation=d, transposed=tr, output_padding=op[0], device=dev)
    say("lav forward")
    t_lf = ev(lambda: lf(x))
    err_f = (lf(x) - y).abs().max().item()
    if tr:
        ld = ConvLayer(w.cpu(), stride=s[0], padding=p, dilation=d, device=dev)
    else:
        oph = xs[2] - ((y.shape[2] - 1) * s[0] - 2 * p[0] + d[0] * (k[0] - 1) + 1)
        ld = ConvLayer(w.cpu(), stride=s[0], padding=p, dilation=d, transposed=True, output_padding=oph, device=dev)
    gx = torch.autograd.grad(yx, xg, dy, retain_graph=True)[0]
    say("lav adjoint")
    try:
        t_ld = ev(lambda: ld(dy))
        err_d = (ld(dy) - gx).abs().max().item()
    except RuntimeError as e:
        t_ld, err_d = float("nan"), float("nan")
        print("   adjoint plan failed:", str(e)[:100])
    for kk, vv in (("fwd", t_f), ("dgrad", t_d), ("wgrad", t_w), ("lav_fwd", t_lf), ("lav_dgrad", t_ld)):
        tot[kk] += vv * cnt
    print(f"x{cnt:3d} {name[:5]:5s} {cin:3d}->{cout:3d} k{k[0]} s{s[0]} {str(xs):24s} torch f/d/w {t_f:8.1f} {t_d:8.1f} {t_w:8.1f} us "
          f"({flops / t_f / 1e6:5.1f} {flops / t_d / 1e6:5.1f} {flops / t_w / 1e6:5.1f} TF/s) | lav f/d {t_lf:8.1f} {t_ld:8.1f} us "
          f"({flops / t_lf / 1e6:5.1f} {flops / t_ld / 1e6:5.1f}) err {err_f:.1e} {err_d:.1e}", flush=True)
print("totals per step (us):", {k: round(v) for k, v in tot.items()})
