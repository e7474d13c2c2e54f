"""ResNet-18 trunk (conv1 .. layer4) with the reference's state_dict keys (lav/models/resnet.py:148-250,
`num_channels` input planes, forward stops after layer4), evaluated with liblav_amd's MFMA convolution:
Conv -> BatchNorm (-> + identity) -> ReLU is one lav_conv2d launch per convolution.
"""
from __future__ import annotations

import torch
from torch import nn
import torch.nn.functional as F

from .lidar import _Engine, _bn_tuple
from .ops import ConvLayer


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
        eng = self._fresh(device)
        if eng is not None:
            return eng
        def cl(conv, bn, **kw):
            return ConvLayer(conv.weight, stride=conv.stride[0], padding=conv.padding, bn=_bn_tuple(bn), bn_eps=bn.eps,
                             device=device, target_cus=getattr(self, "target_cus", 0), **kw)
        blocks = []
        for i in range(1, 5):
            for blk in getattr(self, f"layer{i}"):
                blocks.append(dict(c1=cl(blk.conv1, blk.bn1, relu_post=True), c2=cl(blk.conv2, blk.bn2, relu_post=True),
                                   down=None if blk.downsample is None else cl(blk.downsample[0], blk.downsample[1])))
        eng = dict(device=device, stem=cl(self.conv1, self.bn1, relu_post=True), blocks=blocks)
        eng["tensor_ids"] = self._tensor_ids()   # (recorded where the engine is built: lidar.py:_Engine._fresh)
        object.__setattr__(self, "_eng", eng)
        return eng

    def forward_train(self, x):
        """Train mode (autograd, BatchNorm on batch statistics): plain torch ops over the same modules."""
        from .train.hipnn import bn_act, conv_module as cv   # fused BatchNorm + ReLU (+ identity); convolutions on liblav_amd (round 5)
        x = self.maxpool(bn_act(self.bn1, cv(self.conv1, x), relu_post=True))
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
        x = ops.maxpool3x3s2(e["stem"](x))     # nn.MaxPool2d(3, 2, 1)
        for b in e["blocks"]:
            identity = x if b["down"] is None else b["down"](x)
            x = b["c2"](b["c1"](x), residual=identity)
        return x


def resnet18(pretrained=False, progress=True, **kwargs):
    return ResNet((2, 2, 2, 2), **kwargs)
