"""ERFNet (5-class semantic segmentation of the three 288x256 cameras) with the reference's state_dict keys
(lav/models/erfnet.py).  Not one of the hand-written kernels of this round: it runs on PyTorch-ROCm (MIOpen)
and is listed as the next conv family to move onto lav_conv2d.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F


class DownsamplerBlock(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout - cin, 3, stride=2, padding=1, bias=True)
        self.pool = nn.MaxPool2d(2, stride=2)
        self.bn = nn.BatchNorm2d(cout, eps=1e-3)

    def forward(self, x):
        return F.relu(self.bn(torch.cat([self.conv(x), self.pool(x)], 1)))


class non_bottleneck_1d(nn.Module):  # name kept: it is part of pickled/traced checkpoints' qualified names
    def __init__(self, ch, dropprob, dilated):
        super().__init__()
        d = dilated
        self.conv3x1_1 = nn.Conv2d(ch, ch, (3, 1), padding=(1, 0))
        self.conv1x3_1 = nn.Conv2d(ch, ch, (1, 3), padding=(0, 1))
        self.bn1 = nn.BatchNorm2d(ch, eps=1e-3)
        self.conv3x1_2 = nn.Conv2d(ch, ch, (3, 1), padding=(d, 0), dilation=(d, 1))
        self.conv1x3_2 = nn.Conv2d(ch, ch, (1, 3), padding=(0, d), dilation=(1, d))
        self.bn2 = nn.BatchNorm2d(ch, eps=1e-3)
        self.dropout = nn.Dropout2d(dropprob)

    def forward(self, x):
        y = F.relu(self.bn1(self.conv1x3_1(F.relu(self.conv3x1_1(x)))))
        y = self.bn2(self.conv1x3_2(F.relu(self.conv3x1_2(y))))
        if self.dropout.p != 0:
            y = self.dropout(y)
        return F.relu(y + x)


class UpsamplerBlock(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.conv = nn.ConvTranspose2d(cin, cout, 3, stride=2, padding=1, output_padding=1, bias=True)
        self.bn = nn.BatchNorm2d(cout, eps=1e-3)

    def forward(self, x):
        return F.relu(self.bn(self.conv(x)))


class Encoder(nn.Module):
    def __init__(self, num_classes):
        super().__init__()
        self.initial_block = DownsamplerBlock(3, 16)
        blocks = [DownsamplerBlock(16, 64)] + [non_bottleneck_1d(64, 0.03, 1) for _ in range(5)] + [DownsamplerBlock(64, 128)]
        blocks += [non_bottleneck_1d(128, 0.3, d) for _ in range(2) for d in (2, 4, 8, 16)]
        self.layers = nn.ModuleList(blocks)
        self.output_conv = nn.Conv2d(128, num_classes, 1)

    def forward(self, x, predict=False):
        x = self.initial_block(x)
        for layer in self.layers:
            x = layer(x)
        return self.output_conv(x) if predict else x


class Decoder(nn.Module):
    def __init__(self, num_classes):
        super().__init__()
        self.layers = nn.ModuleList([UpsamplerBlock(128, 64), non_bottleneck_1d(64, 0, 1), non_bottleneck_1d(64, 0, 1),
                                     UpsamplerBlock(64, 16), non_bottleneck_1d(16, 0, 1), non_bottleneck_1d(16, 0, 1)])
        self.output_conv = nn.ConvTranspose2d(16, num_classes, 2, stride=2)

    def forward(self, x):
        for layer in self.layers:
            x = layer(x)
        return self.output_conv(x)


class ERFNet(nn.Module):
    def __init__(self, num_classes):
        super().__init__()
        self.encoder = Encoder(num_classes)
        self.decoder = Decoder(num_classes)

    def forward(self, x):
        return self.decoder(self.encoder(x))
