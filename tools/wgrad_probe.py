"""lav_conv_wgrad (bf16x6 weight gradient, csrc/conv_wgrad.hip) against torch's (MIOpen) weight gradient on the 3x3 shapes (strides 1 and 2) of a
train_full step at BASELINE's batch 32.    python tools/wgrad_probe.py [batch]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lav_amd import _lib  # noqa: E402
from lav_amd.ops import _ptr, _stream, _workspace, check  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda")
lib = _lib.load()


def ev(fn, reps=5):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


only = os.environ.get("WGRAD_ONLY")
for name, b, cin, cout, H, W, S, KS in (("heads 384->256 @160", B, 384, 256, 160, 160, 1, 3), ("backbone 64->64 @160", B, 64, 64, 160, 160, 1, 3),
                                        ("backbone 128->128 @80", B, 128, 128, 80, 80, 1, 3), ("backbone 128->128 @40", B, 128, 128, 40, 40, 1, 3),
                                        ("resnet 64->64 @24 (crops)", 3 * B, 64, 64, 24, 24, 1, 3),
                                        ("backbone 64->64 s2 @320", B, 64, 64, 320, 320, 2, 3), ("backbone 64->128 s2 @160", B, 64, 128, 160, 160, 2, 3),
                                        ("backbone 128->256 s2 @80", B, 128, 256, 80, 80, 2, 3),
                                        ("stem 7x7 s2 384->64 @96 (ego crops)", B, 384, 64, 96, 96, 2, 7), ("stem 7x7 s2 384->64 @96 (others)", 2 * B, 384, 64, 96, 96, 2, 7)):
    if only and only not in name:
        continue
    x = torch.randn((b, cin, H, W), device=dev)
    dy = torch.randn((b, cout, H // S, W // S), device=dev)
    w = torch.randn((cout, cin, KS, KS), device=dev)
    dw = torch.empty_like(w)
    ws = _workspace("conv_wgrad", lib.lav_conv_wgrad_workspace_bytes(b, cin, cout, H, W, KS, S), dev)
    t_l = ev(lambda: check(lib.lav_conv_wgrad(_ptr(x), _ptr(dy), b, cin, cout, H, W, KS, S, _ptr(dw), _ptr(ws), ws.numel(), _stream()), "wgrad"))
    t_t = ev(lambda: torch.ops.aten.convolution_backward(dy, x, w, None, [S, S], [KS // 2, KS // 2], [1, 1], False, [0, 0], 1, [False, True, False]))
    ref = torch.ops.aten.convolution_backward(dy, x, w, None, [S, S], [KS // 2, KS // 2], [1, 1], False, [0, 0], 1, [False, True, False])[1]
    fl = 2.0 * b * (H // S) * (W // S) * cin * cout * KS * KS
    print(f"{name:36s} batch {b:3d}: lav {t_l:8.1f} us ({fl / t_l / 1e6:6.1f} TF/s eq., {6 * fl / t_l / 1e6 / 2500 * 100:4.1f}% of the bf16 roof)  torch {t_t:8.1f} us ({fl / t_t / 1e6:6.1f} TF/s)  "
          f"max |diff| / max |ref| {float((dw - ref).abs().max() / ref.abs().max()):.1e}  workspace {ws.numel() / 1e6:.0f} MB", flush=True)
