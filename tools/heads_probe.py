"""Time of the heads' pieces in isolation (library HIP events, back-to-back launches): fused first convolution, grouped
deconvolution, peak extraction.  LAV_DECONV_UNROLL=2|4|8 python tools/heads_probe.py"""
import ctypes
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402
from lav_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")
lib = _lib.load()
pipe, sds, (lm, up, seg, bra) = bench.build_pipeline(dev, eager=True)
feats = torch.randn((1, 384, 160, 160), device=dev)


def read(name):
    ms, n = ctypes.c_double(), ctypes.c_int()
    lib.lav_profile_read(name.encode(), ctypes.byref(ms), ctypes.byref(n))
    return ms.value / max(n.value, 1) * 1e3, n.value


with torch.no_grad():
    heat, size, ori, bev = lm.heads(feats)
    fn = lambda: (lm.heads(feats), ops.extract_peaks(heat[0], size[0], ori[0], apply_sigmoid=True))
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    lib.lav_profile_enable(400)
    fn(); torch.cuda.synchronize(); lib.lav_profile_reset()
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    print(f"LAV_DECONV_IMPL={os.environ.get('LAV_DECONV_IMPL', 'staged')}: " + ", ".join(f"{k} {read(k)[0]:.1f} us" for k in ("conv2d", "deconv_grouped", "extract_peaks")))
