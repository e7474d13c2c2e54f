#!/usr/bin/env python3
"""Workload for the MFMA-utilisation PMC pass (rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES ...): the frame's
characteristic convolution kernels, a few launches each - the fused heads convolution (tiled k_conv<2,2>), a BEV backbone
layer on the direct kernel (k_conv_direct), the 7x7 stem of the crop embedder and one ERFNet block (k_conv1d_pair)."""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import lav_amd  # noqa: E402
from lav_amd import synth  # noqa: E402
from lav_amd.rgb import RGBSegmentationModel  # noqa: E402

dev = torch.device("cuda:0")
cfg = dict(min_x=-10, max_x=70, min_y=-40, max_y=40, pixels_per_meter=4)
lm = lav_amd.LiDARModel(num_input=16, backbone="cnn", num_features=[64, 64], **cfg)
lm.load_state_dict(synth.seeded_state_dict(lm, prefix="lidar."))
lm.eval().to(dev)
seg = RGBSegmentationModel([4, 6, 7, 10]); seg.load_state_dict(synth.seeded_state_dict(seg, prefix="seg.")); seg.eval().to(dev)
canvas = torch.randn((1, 64, 320, 320), device=dev)
imgs = torch.rand((3, 3, 288, 256), device=dev) * 255
with torch.no_grad():
    for _ in range(4):
        feats = lm.backbone(canvas)
        lm.heads(feats)
        seg(imgs)
# round 4: the 7x7 stride-2 crop stem (384 -> 64 on 96x96 crops) at 1 and 7 crops: the split kernel's tap-pair mode
from lav_amd import _lib  # noqa: E402
from lav_amd.ops import ConvLayer  # noqa: E402
stem = ConvLayer(torch.randn(64, 384, 7, 7) / (384 * 49) ** 0.5, stride=2, padding=(3, 3), relu_post=True, precision=_lib.CONV_BF16X6, device=dev)
for b in (1, 7):
    x = torch.randn(b, 384, 96, 96, device=dev)
    for _ in range(4):
        stem(x)
torch.cuda.synchronize()
