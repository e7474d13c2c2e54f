#!/usr/bin/env python3
"""Split (bf16x6) convolution kernel against the exact fp32 kernels: time and error per layer shape, sweeping the split
kernel's tile shape / tile width / tap group / split-K through LAV_SPLIT_FORCE.

    python tools/split_probe.py [--sweep] [name-substring ...]
"""
import ctypes
import itertools
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lav_amd import _lib
from lav_amd.ops import ConvLayer
from lav_amd._lib import Conv

SHAPES = [
    # name, B, cin, cout, k, stride, pad, dil, transposed, out_pad, H, W
    ("head 384->256 160", 1, 384, 256, (3, 3), 1, (1, 1), (1, 1), False, 0, 160, 160),
    ("bev 64->64 s2 320", 1, 64, 64, (3, 3), 2, (1, 1), (1, 1), False, 0, 320, 320),
    ("bev 64->64 160", 1, 64, 64, (3, 3), 1, (1, 1), (1, 1), False, 0, 160, 160),
    ("bev 64->128 s2 160", 1, 64, 128, (3, 3), 2, (1, 1), (1, 1), False, 0, 160, 160),
    ("bev 128->128 80", 1, 128, 128, (3, 3), 1, (1, 1), (1, 1), False, 0, 80, 80),
    ("bev 128->128 s2 80", 1, 128, 128, (3, 3), 2, (1, 1), (1, 1), False, 0, 80, 80),
    ("bev 128->128 40", 1, 128, 128, (3, 3), 1, (1, 1), (1, 1), False, 0, 40, 40),
    ("up1 convT 1x1", 1, 64, 128, (1, 1), 1, (0, 0), (1, 1), True, 0, 160, 160),
    ("up2 convT 4x4 s2", 1, 128, 128, (4, 4), 2, (1, 1), (1, 1), True, 0, 80, 80),
    ("up3 convT 4x4 s4", 1, 128, 128, (4, 4), 4, (1, 1), (1, 1), True, 2, 40, 40),
    ("stem 7x7 B1", 1, 384, 64, (7, 7), 2, (3, 3), (1, 1), False, 0, 96, 96),
    ("stem 7x7 B4", 4, 384, 64, (7, 7), 2, (3, 3), (1, 1), False, 0, 96, 96),
    ("stem 7x7 B7", 7, 384, 64, (7, 7), 2, (3, 3), (1, 1), False, 0, 96, 96),
    ("stem 7x7 B15", 15, 384, 64, (7, 7), 2, (3, 3), (1, 1), False, 0, 96, 96),
    ("res l1 64 24x24 B1", 1, 64, 64, (3, 3), 1, (1, 1), (1, 1), False, 0, 24, 24),
    ("res l1 64 24x24 B7", 7, 64, 64, (3, 3), 1, (1, 1), (1, 1), False, 0, 24, 24),
    ("res l2 128 12x12 B7", 7, 128, 128, (3, 3), 1, (1, 1), (1, 1), False, 0, 12, 12),
    ("res l3 256 6x6 B7", 7, 256, 256, (3, 3), 1, (1, 1), (1, 1), False, 0, 6, 6),
    ("res l4 512 3x3 B7", 7, 512, 512, (3, 3), 1, (1, 1), (1, 1), False, 0, 3, 3),
    ("brake stem 7x7 288x768", 1, 3, 64, (7, 7), 2, (3, 3), (1, 1), False, 0, 288, 768),
    ("brake l1 64 72x192", 1, 64, 64, (3, 3), 1, (1, 1), (1, 1), False, 0, 72, 192),
    ("brake l2 128 36x96", 1, 128, 128, (3, 3), 1, (1, 1), (1, 1), False, 0, 36, 96),
    ("brake l3 256 18x48", 1, 256, 256, (3, 3), 1, (1, 1), (1, 1), False, 0, 18, 48),
    ("brake l4 512 9x24", 1, 512, 512, (3, 3), 1, (1, 1), (1, 1), False, 0, 9, 24),
    ("erf down 16->48 s2 B3", 3, 16, 48, (3, 3), 2, (1, 1), (1, 1), False, 0, 144, 128),
    ("erf down 64->64 s2 B3", 3, 64, 64, (3, 3), 2, (1, 1), (1, 1), False, 0, 72, 64),
    ("erf up 128->64 B3", 3, 128, 64, (3, 3), 2, (1, 1), (1, 1), True, 1, 36, 32),
]


def timed(lib, layer, x, reps=20):
    lib.lav_profile_enable(reps + 8)
    layer(x); layer(x); torch.cuda.synchronize(); lib.lav_profile_reset()
    for _ in range(reps):
        layer(x)
    torch.cuda.synchronize()
    ms, n = ctypes.c_double(), ctypes.c_int()
    lib.lav_profile_read(b"conv2d", ctypes.byref(ms), ctypes.byref(n))
    lib.lav_profile_enable(0)
    return ms.value / max(n.value, 1) * 1e3


def plan_str(lib, layer, B, H, W):
    desc = Conv.from_buffer_copy(layer.desc); desc.batch, desc.h, desc.w = B, H, W
    info = (ctypes.c_int * 9)()
    if lib.lav_conv_tile_info(ctypes.byref(desc), info):
        return "?"
    if info[0] == -1:
        return f"split {info[1]}x{info[2]}/w{info[3]} tw{info[4]}xth{info[8]} tg{info[7] % 100}{'p' if info[7] >= 100 else ''} ks{info[6]} lds{info[5] // 1024}K"
    if info[0] == 0:
        return f"direct w{info[1]} mc{info[2]} ks{info[6]}"
    return f"tiled {info[0]}x{info[1]} ks{info[6]} lds{info[5] // 1024}K"


def main():
    lib = _lib.load()
    dev = torch.device("cuda")
    args = sys.argv[1:]
    sweep = "--sweep" in args
    sel = [a for a in args if not a.startswith("--")]
    force = [a.split("=", 1)[1] for a in args if a.startswith("--force=")]
    torch.manual_seed(0)
    for name, B, cin, cout, k, s, p, d, tr, op, H, W in SHAPES:
        if sel and not any(x in name for x in sel):
            continue
        w = torch.randn((cin, cout, *k) if tr else (cout, cin, *k)) / (cin * k[0] * k[1]) ** 0.5
        x = torch.randn((B, cin, H, W), device=dev)
        kw = dict(stride=s, padding=p, dilation=d, transposed=tr, output_padding=op, relu_pre=True, device=dev)
        os.environ.pop("LAV_SPLIT_FORCE", None)
        os.environ["LAV_CONV_SPLIT"] = "1"
        exact = ConvLayer(w, precision=_lib.CONV_F32, **kw)
        y0 = exact(x)
        flops = 2.0 * y0.numel() * cin * k[0] * k[1] / (s * s if tr else 1)
        t0 = timed(lib, exact, x)
        print(f"{name:24s} fp32  {t0:8.1f} us {flops / t0 / 1e6:7.1f} TF/s  [{plan_str(lib, exact, B, H, W)}]", flush=True)
        split = ConvLayer(w, precision=_lib.CONV_F16X3 if os.environ.get("LAV_PROBE_PREC") == "f16" else _lib.CONV_BF16X6, **kw)   # (LAV_PROBE_PREC=f16: the frame's precision)
        t1 = timed(lib, split, x)
        err = (split(x) - y0).abs().max().item()
        print(f"{'':24s} auto  {t1:8.1f} us {flops / t1 / 1e6:7.1f} TF/s  [{plan_str(lib, split, B, H, W)}] max|diff| {err:.2e} (|y| max {y0.abs().max().item():.2f})", flush=True)
        os.environ["LAV_CONV_SPLIT"] = "2"
        split._ws_bytes = {}
        for f in force:
            os.environ["LAV_SPLIT_FORCE"] = f
            split._ws_bytes = {}
            t2 = timed(lib, split, x)
            print(f"{'':24s} {f:14s} {t2:8.1f} us {flops / t2 / 1e6:7.1f} TF/s  [{plan_str(lib, split, B, H, W)}]", flush=True)
            os.environ.pop("LAV_SPLIT_FORCE", None)
        if force:
            continue
        if not sweep:
            t2 = timed(lib, split, x)
            err = (split(x) - y0).abs().max().item()
            print(f"{'':24s} split {t2:8.1f} us {flops / t2 / 1e6:7.1f} TF/s  [{plan_str(lib, split, B, H, W)}] max|diff| {err:.2e}", flush=True)
            continue
        results = []
        for (mp, mc, wpx), tw, tg, ks in itertools.product([(2, 2, 4), (1, 2, 4), (1, 1, 4), (2, 2, 2), (1, 2, 2), (1, 1, 2)], [0, 32, 64, 128],
                                                          [0], [1, 2, 4]):
            os.environ["LAV_SPLIT_FORCE"] = f"{mp},{mc},{wpx},{tw},{tg},{ks}"
            split._ws_bytes = {}
            ps = plan_str(lib, split, B, H, W)
            if not ps.startswith("split"):
                continue
            try:
                t = timed(lib, split, x, reps=10)
            except RuntimeError as e:
                print("   ", os.environ["LAV_SPLIT_FORCE"], "failed:", e)
                continue
            results.append((t, ps))
        os.environ.pop("LAV_SPLIT_FORCE", None)
        seen = set()
        for t, ps in sorted(results):
            if ps in seen:
                continue
            seen.add(ps)
            print(f"{'':24s}   {t:8.1f} us {flops / t / 1e6:7.1f} TF/s  [{ps}]")
            if len(seen) >= 6:
                break
        sys.stdout.flush()


if __name__ == "__main__":
    main()
