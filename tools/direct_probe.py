#!/usr/bin/env python3
"""Calibration of the direct small-layer convolution: every shape timed on the tiled plan and on the direct kernel for
each (waves, split-K) it admits (HIP events around back-to-back launches).

    python tools/direct_probe.py [name-substring ...]
"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lav_amd import _lib  # noqa: E402
from lav_amd.ops import ConvLayer  # noqa: E402
from lav_amd._lib import Conv  # noqa: E402

lib = _lib.load()

SHAPES = [
    # name, B, cin, cout, k, stride, pad, H, W
    ("l1 64 24x24", 64, 64, 3, 1, 1, 24, 24),
    ("l2 down 64->128 s2", 64, 128, 3, 2, 1, 24, 24),
    ("l2 1x1 s2", 64, 128, 1, 2, 0, 24, 24),
    ("l2 128 12x12", 128, 128, 3, 1, 1, 12, 12),
    ("l3 down 128->256 s2", 128, 256, 3, 2, 1, 12, 12),
    ("l3 1x1 s2", 128, 256, 1, 2, 0, 12, 12),
    ("l3 256 6x6", 256, 256, 3, 1, 1, 6, 6),
    ("l4 down 256->512 s2", 256, 512, 3, 2, 1, 6, 6),
    ("l4 1x1 s2", 256, 512, 1, 2, 0, 6, 6),
    ("l4 512 3x3", 512, 512, 3, 1, 1, 3, 3),
    ("stem 7x7 384->64 96", 384, 64, 7, 2, 3, 96, 96),
    ("bev 128 40x40", 128, 128, 3, 1, 1, 40, 40),
    ("bev 128 80x80", 128, 128, 3, 1, 1, 80, 80),
    ("brake l2 128 36x96", 128, 128, 3, 1, 1, 36, 96),
    ("brake l3 256 18x48", 256, 256, 3, 1, 1, 18, 48),
    ("brake l4 512 9x24", 512, 512, 3, 1, 1, 9, 24),
    ("bev 64 160x160", 64, 64, 3, 1, 1, 160, 160),
    ("bev 64->128 s2 160", 64, 128, 3, 2, 1, 160, 160),
    ("brake l1 64 72x192", 64, 64, 3, 1, 1, 72, 192),
    ("head 64->64 1x1 160", 64, 64, 1, 1, 0, 160, 160),
    # transposed (name starts with T): k, stride, pad as for ConvTranspose2d, output_padding 1 for the 3x3 ones
    ("T bev 1x1 64->128 160", 64, 128, 1, 1, 0, 160, 160),
    ("T bev 4x4 s2 128 80", 128, 128, 4, 2, 1, 80, 80),
    ("T bev 4x4 s4 128 40", 128, 128, 4, 4, 0, 40, 40),
    ("T erf 3x3 s2 128->64", 128, 64, 3, 2, 1, 36, 32),
    ("T erf 3x3 s2 64->16", 64, 16, 3, 2, 1, 72, 64),
]
dev = torch.device("cuda")


def timed(layer, x, reps=20):
    """Mean of the library's own HIP-event bracket around each launch (kernel + split-K reduce), not Python's pace."""
    layer(x)
    lib.lav_profile_enable(reps + 8)
    layer(x); torch.cuda.synchronize(); lib.lav_profile_reset()
    for _ in range(reps):
        layer(x)
    torch.cuda.synchronize()
    ms, n = ctypes.c_double(), ctypes.c_int()
    lib.lav_profile_read(b"conv2d", ctypes.byref(ms), ctypes.byref(n))
    lib.lav_profile_enable(0)
    return ms.value / max(n.value, 1) * 1e3


def main():
    sel = sys.argv[1:]
    batches = [int(b) for b in os.environ.get("BATCHES", "1,7").split(",")]
    for name, cin, cout, k, s, p, H, W in SHAPES:
        if sel and not any(t in name for t in sel):
            continue
        tr = name.startswith("T ")
        w = torch.randn((cin, cout, k, k) if tr else (cout, cin, k, k)) * 0.05
        for B in batches:
            x = torch.randn((B, cin, H, W), device=dev)
            res = {}
            for cfg in ["tiled", "auto"] + [f"w{wv}k{ks}m{mc}" for mc in (1, 2) for wv in (4, 8, 16) for ks in (1, 2, 4, 8, 16)]:
                os.environ.pop("LAV_CONV_DIRECT_WAVES", None); os.environ.pop("LAV_CONV_DIRECT_KS", None); os.environ.pop("LAV_CONV_DIRECT_MC", None)
                if cfg == "tiled":
                    os.environ["LAV_CONV_DIRECT"] = "0"
                elif cfg == "auto":
                    os.environ["LAV_CONV_DIRECT"] = "1"
                else:
                    wv, rest = cfg[1:].split("k")
                    ks, mc = rest.split("m")
                    if cin % (8 * int(wv) * int(ks)) or (mc == "2" and (cout < 64 or int(wv) > 8)):
                        continue
                    os.environ.update(LAV_CONV_DIRECT="2", LAV_CONV_DIRECT_WAVES=wv, LAV_CONV_DIRECT_KS=ks, LAV_CONV_DIRECT_MC=mc)
                layer = ConvLayer(w, stride=s, padding=p, relu_post=True, transposed=tr, output_padding=1 if (tr and k == 3) else 0, device=dev)
                try:
                    res[cfg] = timed(layer, x)
                except RuntimeError as e:
                    res[cfg] = float("nan")
            best = min((v, c) for c, v in res.items() if c not in ("tiled", "auto") and v == v)
            d = Conv.from_buffer_copy(layer.desc); d.batch, d.h, d.w = B, H, W
            info = (ctypes.c_int * 9)()
            os.environ["LAV_CONV_DIRECT"] = "1"; os.environ.pop("LAV_CONV_DIRECT_WAVES", None); os.environ.pop("LAV_CONV_DIRECT_KS", None); os.environ.pop("LAV_CONV_DIRECT_MC", None)
            lib.lav_conv_tile_info(ctypes.byref(d), info)
            plan = f"w{info[1]}k{info[6]}m{info[2]}" if info[0] == 0 else f"tile{info[0]}x{info[1]}k{info[6]}"
            print(f"{name:24s} B={B}  tiled {res['tiled']:6.1f}  auto {res['auto']:6.1f} ({plan:10s})  best direct {best[1]:6s} {best[0]:6.1f} | " +
                  " ".join(f"{c}:{v:.1f}" for c, v in sorted(res.items(), key=lambda cv: cv[1])[:8] if c not in ("tiled", "auto")), flush=True)


if __name__ == "__main__":
    main()
