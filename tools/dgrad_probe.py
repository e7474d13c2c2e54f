"""Data gradient of the training graph's convolutions: the transposed lav_conv2d plan that lav_amd/train/hipnn.py launches against torch's
(MIOpen) data gradient, per layer shape at BASELINE's batch 32.    python tools/dgrad_probe.py [batch]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lav_amd.train.hipnn import _conv_engine  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda")


def ev(fn, reps=5):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


only = os.environ.get("DGRAD_ONLY")
for name, b, cin, cout, k, s, H, W in (("stem 7x7 s2 384->64 @96 (ego)", B, 384, 64, 7, 2, 96, 96), ("stem 7x7 s2 384->64 @96 (others)", 3 * B, 384, 64, 7, 2, 96, 96),
                                       ("backbone 64->64 s2 @320", B, 64, 64, 3, 2, 320, 320), ("backbone 64->128 s2 @160", B, 64, 128, 3, 2, 160, 160),
                                       ("backbone 128->256 s2 @80", B, 128, 256, 3, 2, 80, 80), ("resnet 64->128 s2 @24", 3 * B, 64, 128, 3, 2, 24, 24),
                                       ("resnet 128->256 s2 @12", 3 * B, 128, 256, 3, 2, 12, 12), ("resnet 256->512 s2 @6", 3 * B, 256, 512, 3, 2, 6, 6),
                                       ("heads 384->256 @160", B, 384, 256, 3, 1, 160, 160), ("backbone 64->64 @160", B, 64, 64, 3, 1, 160, 160),
                                       ("backbone 128->128 @80", B, 128, 128, 3, 1, 80, 80), ("backbone 256->256 @40", B, 256, 256, 3, 1, 40, 40),
                                       ("resnet 64->64 @24", 3 * B, 64, 64, 3, 1, 24, 24)):
    if only and only not in name:
        continue
    p = k // 2
    OH, OW = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    x = torch.randn((b, cin, H, W), device=dev)
    dy = torch.randn((b, cout, OH, OW), device=dev)
    w = torch.randn((cout, cin, k, k), device=dev) / (cin * k * k) ** 0.5
    oph = H - ((OH - 1) * s - 2 * p + (k - 1) + 1)
    eng = _conv_engine("dgrad", w, s, (p, p), (1, 1), True, oph)
    t_l = ev(lambda: eng(dy))
    t_t = ev(lambda: torch.ops.aten.convolution_backward(dy, x, w, None, [s, s], [p, p], [1, 1], False, [0, 0], 1, [True, False, False]))
    ref = torch.ops.aten.convolution_backward(dy, x, w, None, [s, s], [p, p], [1, 1], False, [0, 0], 1, [True, False, False])[0]
    got = eng(dy)
    fl = 2.0 * b * OH * OW * cin * cout * k * k
    print(f"{name:36s} batch {b:3d}: lav {t_l:8.1f} us ({fl / t_l / 1e6:6.1f} TF/s eq.)  torch {t_t:8.1f} us ({fl / t_t / 1e6:6.1f} TF/s)  "
          f"max |diff| / max |ref| {float((got - ref).abs().max() / ref.abs().max()):.1e}", flush=True)
