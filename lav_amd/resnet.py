"""ResNet-18 trunk (conv1 .. layer4) with the reference's state_dict keys (lav/models/resnet.py:148-250,
`num_channels` input planes, forward stops after layer4), evaluated with liblav_amd's MFMA convolution:
Conv -> BatchNorm (-> + identity) -> ReLU is one lav_conv2d launch per convolution.
"""
from __future__ import annotations

import torch
from torch import nn
import torch.nn.functional as F

from . import _lib
from .lidar import _Engine, _bn_tuple
from .ops import Amax, ConvLayer


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = None
        if stride != 1 or inplanes != planes:
            self.downsample = nn.Sequential(nn.Conv2d(inplanes, planes, 1, stride, bias=False), nn.BatchNorm2d(planes))
        self.stride = stride


class ResNet(_Engine):
    def __init__(self, layers=(2, 2, 2, 2), num_channels=3, num_classes=1000):
        super().__init__()
        self.conv1 = nn.Conv2d(num_channels, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        planes, inpl = (64, 128, 256, 512), 64
        for i, (p, n) in enumerate(zip(planes, layers)):
            blocks = []
            for j in range(n):
                blocks.append(BasicBlock(inpl, p, (1 if i == 0 else 2) if j == 0 else 1))
                inpl = p
            setattr(self, f"layer{i + 1}", nn.Sequential(*blocks))
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512, num_classes)  # present in the reference's checkpoints, unused by forward
        self._drop()

    def _engine(self, device):
        """The packed layers for the calling thread's inference precision (ops.infer_precision), one engine per precision."""
        from . import ops
        prec = ops.infer_precision()
        eng = self._fresh(device) or dict(device=device)
        key = ("trunk", prec)
        if key in eng:
            return eng[key]
        def cl(conv, bn, **kw):
            return ConvLayer(conv.weight, stride=conv.stride[0], padding=conv.padding, bn=_bn_tuple(bn), bn_eps=bn.eps,
                             device=device, target_cus=getattr(self, "target_cus", 0), precision=prec, **kw)
        blocks = []
        for i in range(1, 5):
            for blk in getattr(self, f"layer{i}"):
                blocks.append(dict(c1=cl(blk.conv1, blk.bn1, relu_post=True), c2=cl(blk.conv2, blk.bn2, relu_post=True),
                                   down=None if blk.downsample is None else cl(blk.downsample[0], blk.downsample[1])))
        eng[key] = dict(stem=cl(self.conv1, self.bn1, relu_post=True), blocks=blocks, amax={}, f16=prec == _lib.CONV_F16X3)
        eng["tensor_ids"] = self._tensor_ids()   # (recorded where the engine is built: lidar.py:_Engine._fresh)
        object.__setattr__(self, "_eng", eng)
        return eng[key]

    def forward_train(self, x):
        """Train mode (autograd, BatchNorm on batch statistics): plain torch ops over the same modules."""
        from .train.hipnn import bn_act, carry, conv_module as cv   # fused BatchNorm + ReLU (+ identity); convolutions on liblav_amd (round 5)
        y = bn_act(self.bn1, cv(self.conv1, x), relu_post=True)
        x = carry(self.maxpool(y), y)
        for i in range(1, 5):
            for blk in getattr(self, f"layer{i}"):
                identity = x if blk.downsample is None else bn_act(blk.downsample[1], blk.downsample[0](x))
                x = bn_act(blk.bn2, cv(blk.conv2, bn_act(blk.bn1, cv(blk.conv1, x), relu_post=True)), relu_post=True, residual=identity)
        return x

    def forward(self, x):
        if self.training:
            return self.forward_train(x)
        e = self._engine(x.device)
        from . import ops
        if not e["f16"]:
            x = ops.maxpool3x3s2(e["stem"](x))     # nn.MaxPool2d(3, 2, 1)
            for b in e["blocks"]:
                identity = x if b["down"] is None else b["down"](x)
                x = b["c2"](b["c1"](x), residual=identity)
            return x
        # LAV_CONV_F16X3: every convolution leaves the maxima of its output for the layers that read it (lav_conv2d_amax); the stem
        # takes those of its input from whoever made it (the crops carry the feature map's), the max-pool passes its input's on
        B = x.shape[0]
        ams = e["amax"].get(tuple(x.shape))
        if ams is None:
            ams = e["amax"][tuple(x.shape)] = [Amax(x.device) for _ in range(1 + 2 * len(e["blocks"]))]
        blocks = e["blocks"]
        def want(readers, h, w):
            return any(r is not None and r.uses_amax(B, h, w) for r in readers)
        oh, ow = e["stem"].out_hw(x.shape[2], x.shape[3])
        ph, pw = (oh - 1) // 2 + 1, (ow - 1) // 2 + 1
        am_x = ams[0].reset() if want([blocks[0]["c1"], blocks[0]["down"]], ph, pw) else None
        x = ops.maxpool3x3s2(e["stem"](x, amax_in=ops.amax_of(x), amax_out=am_x))
        for i, b in enumerate(blocks):
            identity = x if b["down"] is None else b["down"](x, amax_in=am_x)
            h1, w1 = b["c1"].out_hw(x.shape[2], x.shape[3])
            am_h = ams[1 + 2 * i].reset() if want([b["c2"]], h1, w1) else None
            h = b["c1"](x, amax_in=am_x, amax_out=am_h)
            nxt = blocks[i + 1] if i + 1 < len(blocks) else None
            am_y = ams[2 + 2 * i].reset() if nxt is not None and want([nxt["c1"], nxt["down"]], h1, w1) else None
            x = b["c2"](h, residual=identity, amax_in=am_h, amax_out=am_y)
            am_x = am_y
        return x


def resnet18(pretrained=False, progress=True, **kwargs):
    return ResNet((2, 2, 2, 2), **kwargs)
