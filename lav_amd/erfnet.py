"""ERFNet (5-class semantic segmentation of the three 288x256 cameras) with the reference's state_dict keys
(lav/models/erfnet.py).  In eval mode on the GPU every convolution - 3x3/s2 downsamplers, the factorised
3x1 / 1x3 (dilated) pairs, the 3x3/s2 and 2x2/s2 transposed convolutions - is one lav_conv2d launch with its
bias, BatchNorm affine, residual add and ReLU fused into the epilogue (the reference issues ~6 kernels per
non_bottleneck_1d convolution pair; MIOpen falls back to naive kernels for the dilated asymmetric shapes).
The torch forward below is kept for training mode.
"""
from __future__ import annotations

import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .ops import Conv1dPair, Conv1dPairChain, ConvLayer, GroupedDeconv, infer_precision

_USE_PAIRS = os.environ.get("LAV_ERFNET_PAIRS", "1") != "0"   # A/B switch: 0 = four lav_conv2d launches per block


def _affine(bn, sl):
    s = (bn.weight.detach().double() / torch.sqrt(bn.running_var.detach().double() + bn.eps))
    t = bn.bias.detach().double() - bn.running_mean.detach().double() * s
    return s[sl].float()[None, :, None, None].contiguous(), t[sl].float()[None, :, None, None].contiguous()


class DownsamplerBlock(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout - cin, 3, stride=2, padding=1, bias=True)
        self.pool = nn.MaxPool2d(2, stride=2)
        self.bn = nn.BatchNorm2d(cout, eps=1e-3)

    def forward(self, x):
        return F.relu(self.bn(torch.cat([self.conv(x), self.pool(x)], 1)))

    def engine(self, device, input_affine=None):
        """Two launches: the convolution branch (bias, BatchNorm slice, ReLU fused) and lav_pool_affine for the pooled
        branch, each writing its channel window of the output.  input_affine=(s, t): the block sees RAW input x while the
        reference feeds x' = s*x + t (s > 0): the affine is folded into the convolution (W*s, b + t*sum(W), out-of-image
        taps read -t/s, the pre-image of the reference's zero padding) and into the pooled branch's BatchNorm
        (max commutes with an increasing map)."""
        from . import ops
        nconv = self.conv.out_channels
        cout = self.bn.num_features
        w, b, pad = self.conv.weight.detach(), self.conv.bias.detach(), 0.0
        s_in, t_in = (1.0, 0.0) if input_affine is None else input_affine
        if input_affine is not None:
            if not s_in > 0:
                raise RuntimeError("input_affine needs a positive scale")
            b = (b.double() + t_in * w.double().sum(dim=(1, 2, 3))).float()
            w = (w.double() * s_in).float()
            pad = -t_in / s_in
        sl = slice(0, nconv)
        bn = tuple(t[sl] for t in (self.bn.running_mean, self.bn.running_var, self.bn.weight, self.bn.bias))
        conv = ConvLayer(w, stride=2, padding=(1, 1), bias=b, bn=bn, bn_eps=self.bn.eps, relu_post=True, out_c_total=cout,
                         pad_value=pad, device=device)
        ps, pt = _affine(self.bn, slice(nconv, cout))
        ps, pt = ps.reshape(-1).double(), pt.reshape(-1).double()
        ps, pt = (ps * s_in).float().to(device), (pt + ps * t_in).float().to(device)

        def run(x):
            out = conv(x)                                                    # channels [0, nconv): conv+bias -> BN -> ReLU
            return ops.pool_affine(x, ps, pt, out, nconv, relu=True)         # channels [nconv, cout): pool -> BN -> ReLU
        return run


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

    def engine(self, device):
        a = ConvLayer.from_module(self.conv3x1_1, relu_post=True, device=device)
        b = ConvLayer.from_module(self.conv1x3_1, self.bn1, relu_post=True, device=device)
        c = ConvLayer.from_module(self.conv3x1_2, relu_post=True, device=device)
        d = ConvLayer.from_module(self.conv1x3_2, self.bn2, relu_post=True, device=device)   # (+x) then ReLU
        # each 3x1 -> 1x3 pair as ONE launch (row-tile kernel, intermediate in LDS) where the shape allows it
        p1 = Conv1dPair(self.conv3x1_1, self.conv1x3_1, self.bn1, device=device)
        p2 = Conv1dPair(self.conv3x1_2, self.conv1x3_2, self.bn2, device=device)

        def run(x):
            if _USE_PAIRS and p1.supported(x) and p2.supported(x):
                return p2(p1(x), residual=x)
            return d(c(b(a(x))), residual=x)
        run.pairs = (p1, p2)     # (ERFNet._engine chains the pairs of consecutive blocks into one persistent launch)
        return run


def _chain_blocks(stages):
    """Consecutive non_bottleneck_1d engines -> one Conv1dPairChain each (lav_conv1d_pair_chain: a persistent launch per run of
    blocks - 10 pairs at 64 channels, 16 at 128, 4 + 4 in the decoder), falling back to the blocks' own launches for shapes the
    run does not take (too many rows for the chip, exact-fp32 precision)."""
    out, run = [], []

    def flush():
        if not run:
            return
        blocks = list(run)
        run.clear()
        if len(blocks) == 1 or not _USE_PAIRS:
            out.extend(blocks)
            return
        pairs = [p for b in blocks for p in b.pairs]
        chain = Conv1dPairChain(pairs, [i % 2 == 1 for i in range(len(pairs))])

        def go(x):
            if chain.supported(x):
                return chain(x)
            for b in blocks:
                x = b(x)
            return x
        go.chain = chain
        out.append(go)

    for st in stages:
        if getattr(st, "pairs", None) is not None and (not run or run[-1].pairs[0].ch == st.pairs[0].ch):
            run.append(st)
        else:
            flush()
            if getattr(st, "pairs", None) is not None:
                run.append(st)
            else:
                out.append(st)
    flush()
    return out


class UpsamplerBlock(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.conv = nn.ConvTranspose2d(cin, cout, 3, stride=2, padding=1, output_padding=1, bias=True)
        self.bn = nn.BatchNorm2d(cout, eps=1e-3)

    def forward(self, x):
        return F.relu(self.bn(self.conv(x)))

    def engine(self, device):
        return ConvLayer.from_module(self.conv, self.bn, relu_post=True, device=device)


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
        object.__setattr__(self, "_eng", None)

    def _drop(self):
        object.__setattr__(self, "_eng", None)

    def _apply(self, fn, *a, **k):
        self._drop()
        return super()._apply(fn, *a, **k)

    def train(self, mode: bool = True):
        self._drop()
        return super().train(mode)

    def _load_from_state_dict(self, *a, **k):
        self._drop()
        return super()._load_from_state_dict(*a, **k)

    def _engine(self, device, input_affine=None, softmax=False):
        key = (device, input_affine, softmax, infer_precision())   # (the runs of blocks are built for the precision in force: fp16 pieces under LAV_CONV_F16X3)
        if self._eng is None or self._eng[0] != key:
            stages = [self.encoder.initial_block.engine(device, input_affine)]
            stages += [m.engine(device) for m in self.encoder.layers]
            stages += [m.engine(device) for m in self.decoder.layers]
            stages = _chain_blocks(stages)
            stages.append(GroupedDeconv([self.decoder.output_conv], softmax=softmax, device=device))   # 16 -> classes, k2 s2: memory bound
            object.__setattr__(self, "_eng", (key, stages))
        return self._eng[1]

    def forward(self, x, input_affine=None, softmax=False):
        """input_affine=(s, t): x is the raw input and the network behaves as on s*x + t (folded into the first block in
        the HIP path, applied explicitly in the torch path).  softmax: return the class probabilities instead of the scores
        (computed in the output layer's epilogue in the HIP path)."""
        if self.training:      # autograd path (torch ops)
            if input_affine is not None:
                x = x * input_affine[0] + input_affine[1]
            y = self.decoder(self.encoder(x))
            return torch.softmax(y, dim=1) if softmax else y
        if not x.is_cuda:
            raise RuntimeError("ERFNet: eval-mode forward needs a tensor in HBM - lav_amd has no CPU path "
                               "(CPU evaluation for tests and baselines: oracle/camera.py)")
        # round 6: the persistent runs of one pass share ONE cleaning of their progress counters - the first run zeroes the whole counter
        # array, each run keeps its counters in a region of its own behind the previous run's (ops.pair_chain_region) - instead of a small
        # launch in front of every run (4 x 4.7 us of the frame's lidar graph).  LAV_CHAIN_SHARED_CLEAN=0: every run cleans its own.
        shared = os.environ.get("LAV_CHAIN_SHARED_CLEAN", "1") != "0"
        cap, off, first = ops.pair_chain_capacity(x.device), 0, True
        for stage in self._engine(x.device, input_affine, softmax):
            chain = getattr(stage, "chain", None)
            rows = x.shape[0] * x.shape[2]
            if shared and chain is not None and chain.supported(x) and off + rows <= cap:
                with ops.pair_chain_region(off, cap if first else 0):
                    x = stage(x)
                off, first = off + rows, False
            else:
                if chain is not None:
                    shared = False   # (a run outside the scheme cleans rows from 0 on: later runs must not rely on the shared cleaning)
                x = stage(x)
        return x
