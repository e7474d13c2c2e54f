#!/usr/bin/env python3
"""Time individual lav_conv2d shapes of the frame (back-to-back launches, HIP events from the library).

    python tools/conv_probe.py [name-substring ...]
"""
import ctypes
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lav_amd import _lib
from lav_amd.ops import ConvLayer
from lav_amd._lib import Conv

SHAPES = [
    # name, B, cin, cout, k, stride, pad, dil, transposed, out_pad, H, W
    ("bev 64->64 s2 320", 1, 64, 64, (3, 3), 2, (1, 1), (1, 1), False, 0, 320, 320),
    ("bev 64->64 160", 1, 64, 64, (3, 3), 1, (1, 1), (1, 1), False, 0, 160, 160),
    ("bev 128->128 80", 1, 128, 128, (3, 3), 1, (1, 1), (1, 1), False, 0, 80, 80),
    ("bev 128->128 40", 1, 128, 128, (3, 3), 1, (1, 1), (1, 1), False, 0, 40, 40),
    ("head 384->256 160", 1, 384, 256, (3, 3), 1, (1, 1), (1, 1), False, 0, 160, 160),
    ("up2 convT 4x4 s2", 1, 128, 128, (4, 4), 2, (1, 1), (1, 1), True, 0, 80, 80),
    ("resnet stem 7x7 B1", 1, 384, 64, (7, 7), 2, (3, 3), (1, 1), False, 0, 96, 96),
    ("resnet stem 7x7 B4", 4, 384, 64, (7, 7), 2, (3, 3), (1, 1), False, 0, 96, 96),
    ("resnet l1 64 24x24", 1, 64, 64, (3, 3), 1, (1, 1), (1, 1), False, 0, 24, 24),
    ("resnet l4 512 3x3", 1, 512, 512, (3, 3), 1, (1, 1), (1, 1), False, 0, 3, 3),
    ("brake stem 7x7 288x768", 1, 3, 64, (7, 7), 2, (3, 3), (1, 1), False, 0, 288, 768),
    ("erf 128 3x1 36x32 B3", 3, 128, 128, (3, 1), 1, (1, 0), (1, 1), False, 0, 36, 32),
    ("erf 64 1x3 72x64 B3", 3, 64, 64, (1, 3), 1, (0, 1), (1, 1), False, 0, 72, 64),
    ("erf 16 3x1 144x128 B3", 3, 16, 16, (3, 1), 1, (1, 0), (1, 1), False, 0, 144, 128),
    ("tiny 16 3x1 8x16 (1 WG)", 1, 16, 16, (3, 1), 1, (1, 0), (1, 1), False, 0, 8, 16),
    ("tiny 64 3x3 8x16 (2 WG)", 1, 64, 64, (3, 3), 1, (1, 1), (1, 1), False, 0, 8, 16),
    ("tiny 16 1x1 8x16 (1 WG)", 1, 16, 16, (1, 1), 1, (0, 0), (1, 1), False, 0, 8, 16),
]


def main():
    lib = _lib.load()
    dev = torch.device("cuda")
    sel = sys.argv[1:]
    for name, B, cin, cout, k, s, p, d, tr, op, H, W in SHAPES:
        if sel and not any(x in name for x in sel):
            continue
        w = torch.randn((cin, cout, *k) if tr else (cout, cin, *k)) * 0.05
        layer = ConvLayer(w, stride=s, padding=p, dilation=d, transposed=tr, output_padding=op, relu_pre=True, device=dev)
        x = torch.randn((B, cin, H, W), device=dev)
        desc = Conv.from_buffer_copy(layer.desc); desc.batch, desc.h, desc.w = B, H, W
        info = (ctypes.c_int * 9)()
        lib.lav_conv_tile_info(ctypes.byref(desc), info)
        y = layer(x)
        reps = 30
        lib.lav_profile_enable(reps + 8)
        layer(x); torch.cuda.synchronize(); lib.lav_profile_reset()
        for _ in range(reps):
            layer(x)
        torch.cuda.synchronize()
        ms, n = ctypes.c_double(), ctypes.c_int()
        lib.lav_profile_read(b"conv2d", ctypes.byref(ms), ctypes.byref(n))
        lib.lav_profile_enable(0)
        us = ms.value / max(n.value, 1) * 1e3
        flops = 2.0 * y.numel() * cin * k[0] * k[1] / (s * s if tr else 1)
        print(f"{name:28s} {us:8.1f} us  {flops / us / 1e6:7.1f} TF/s  tile MPxMC={info[0]}x{info[1]} rowblock={info[2]} Wst={info[3]} ROWS={info[4]} "
              f"lds={info[5] // 1024}KB ksplit={info[6]} tapgroup={info[7]} cps={info[8]}")


if __name__ == "__main__":
    main()
