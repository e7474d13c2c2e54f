"""Camera-side models with the reference's state_dict keys (team_code_v2/models/rgb.py:36-83,
lav/models/attention.py, lav/models/segmentation.py): the ERFNet segmenter that feeds point painting and the
brake predictor.  Their convolutions run on liblav_amd's MFMA convolution in eval mode on the GPU (the brake net is called every
frame by the agent, lav_agent_fast.py:323), the single-query attention pooling on lav_attn_pool.
"""
from __future__ import annotations

import math

import torch
from torch import nn
import torch.nn.functional as F

from .erfnet import ERFNet
from . import ops
from . import resnet as _hip_resnet


class Normalize(nn.Module):
    def __init__(self, mean, std):
        super().__init__()
        self.mean = nn.Parameter(torch.tensor(mean), requires_grad=False)
        self.std = nn.Parameter(torch.tensor(std), requires_grad=False)

    def forward(self, x):
        return (x - self.mean[None, :, None, None]) / self.std[None, :, None, None]


class RGBSegmentationModel(nn.Module):
    def __init__(self, seg_channels):
        super().__init__()
        self.erfnet = ERFNet(len(seg_channels) + 1)

    def forward(self, rgb, softmax=False):
        if not self.training:     # (rgb/255 - .5)*2 folded into the first block: three launches less
            return self.erfnet(rgb, input_affine=(2.0 / 255.0, -1.0), softmax=softmax)
        return self.erfnet((rgb / 255. - .5) * 2, softmax=softmax)

    def probs(self, rgb):
        """softmax(forward(rgb), dim=1) - what the agent feeds to point painting (lav_agent_fast.py:264) - with the softmax in
        the output layer's epilogue."""
        return self.forward(rgb, softmax=True)


class SegmentationHead(nn.Module):
    def __init__(self, input_channels, num_labels):
        super().__init__()
        chans = [input_channels, 256, 128, 64]
        mods = []
        for a, b in zip(chans[:-1], chans[1:]):
            mods += [nn.ConvTranspose2d(a, b, 3, 2, 1, 1), nn.BatchNorm2d(b), nn.ReLU(True)]
        mods.append(nn.Conv2d(64, num_labels, 1, 1, 0))
        self.upconv = nn.Sequential(*mods)

    def forward(self, x):
        return self.upconv(x)


_PE_CACHE = {}


def _pe_on(device, d_model, length):
    """Positional encoding resident on `device` (built once: no host->device copy per frame, graph capturable)."""
    key = (str(device), d_model, length)
    if key not in _PE_CACHE:
        _PE_CACHE[key] = positionalencoding1d(d_model, length).to(device)
    return _PE_CACHE[key]


def positionalencoding1d(d_model, length):
    pe = torch.zeros(length, d_model)
    pos = torch.arange(0, length).unsqueeze(1).float()
    div = torch.exp(torch.arange(0, d_model, 2, dtype=torch.float) * -(math.log(10000.0) / d_model))
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    return pe


class Attention(nn.Module):
    """Single learned query per head pooling the (h*w) tokens of the ResNet map (lav/models/attention.py:6-38).
    Eval mode on the GPU: one liblav_amd launch (lav_attn_pool) on projections folded around the query - prepared from
    the parameters once per (device, token count) and rebuilt when a parameter changes; train mode: the reference's
    torch ops (autograd)."""

    def __init__(self, dim, num_heads=8):
        super().__init__()
        self.num_heads, self.dim_head = num_heads, dim // num_heads
        self.q = nn.Parameter(torch.randn(1, num_heads, 1, self.dim_head))
        self.linear_kv = nn.Linear(dim, dim * 2)
        self.scale = self.dim_head ** -0.5
        self._folded = {}

    def _fold(self, device, n_tokens):
        ver = (self.q._version, self.linear_kv.weight._version, self.linear_kv.bias._version,
               self.q.data_ptr(), self.linear_kv.weight.data_ptr())
        key = (str(device), n_tokens)
        hit = self._folded.get(key)
        if hit is not None and hit[0] == ver:
            return hit[1]
        C, H, dh = self.linear_kv.in_features, self.num_heads, self.dim_head
        W = self.linear_kv.weight.detach().double().cpu()
        b = self.linear_kv.bias.detach().double().cpu()
        q = self.q.detach().double().cpu().view(H, dh)
        wk, bk = W[:C].view(H, dh, C), b[:C].view(H, dh)
        u = self.scale * torch.einsum("hd,hdc->hc", q, wk)                                   # (heads, C)
        pe = positionalencoding1d(dh, n_tokens).double()                                     # (N, dh), shared by the heads
        dots_bias = self.scale * (torch.einsum("hd,hd->h", q, bk)[:, None] + q @ pe.t())     # (heads, N)
        f32 = lambda t: t.float().contiguous().to(device)
        folded = (f32(u), f32(dots_bias), f32(W[C:]), f32(b[C:]))
        self._folded[key] = (ver, folded)
        return folded

    def forward(self, x, out=None):
        b, d, h, w = x.shape
        if not self.training:
            if not x.is_cuda:
                raise RuntimeError("Attention: eval-mode forward needs a tensor in HBM - lav_amd has no CPU path (oracle/camera.py)")
            u, dots_bias, w_v, b_v = self._fold(x.device, h * w)
            return ops.attn_pool(x, u, dots_bias, w_v, b_v, self.num_heads, out=out)
        tok = x.flatten(2).transpose(1, 2)                                        # b (h w) d
        k, v = self.linear_kv(tok).chunk(2, dim=-1)
        k = k.view(b, h * w, self.num_heads, self.dim_head).transpose(1, 2)       # b heads n dh
        v = v.view(b, h * w, self.num_heads, self.dim_head).transpose(1, 2)
        k = k + _pe_on(k.device, self.dim_head, h * w)
        attn = torch.softmax(torch.matmul(self.q.expand(b, -1, -1, -1), k.transpose(-1, -2)) * self.scale, dim=-1)
        return torch.matmul(attn, v).transpose(1, 2).reshape(b, d)


class _ResNet18(_hip_resnet.ResNet):
    """ResNet-18 trunk of the brake net: lav_amd.resnet.ResNet (MFMA convolutions) in eval mode, torch ops (autograd) in
    train mode.  Eval mode has no CPU path."""

    def __init__(self, num_channels=3):
        super().__init__((2, 2, 2, 2), num_channels=num_channels)

    def forward(self, x):
        if not self.training:
            if not x.is_cuda:
                raise RuntimeError("brake ResNet-18: eval-mode forward needs a tensor in HBM - lav_amd has no CPU path (oracle/camera.py)")
            return super().forward(x)
        x = self.maxpool(F.relu(self.bn1(self.conv1(x))))
        for i in range(1, 5):
            for blk in getattr(self, f"layer{i}"):
                idt = x if blk.downsample is None else blk.downsample(x)
                x = F.relu(blk.bn2(blk.conv2(F.relu(blk.bn1(blk.conv1(x))))) + idt)
        return x


class RGBBrakePredictionModel(nn.Module):
    def __init__(self, seg_channels, pretrained=False):
        super().__init__()
        self.conv_backbone = _ResNet18(3)
        self.normalize = Normalize(mean=[0.485, 0.456, 0.406], std=[0.229, 0.224, 0.225])
        self.seg_head = SegmentationHead(512, len(seg_channels) + 1)
        self.attn1 = Attention(512, num_heads=8)
        self.attn2 = Attention(512, num_heads=8)
        self.classifier = nn.Sequential(nn.Linear(1024, 1), nn.Sigmoid())

    def trunk(self, rgb):
        """One image through normalize + the shared ResNet-18 (eval HIP path when the image allows it) - forward's first half, also
        called by the frame pipeline when it runs the two images' trunks as separate graphs."""
        if not self.training and rgb.is_cuda and rgb.dtype == torch.float32 and rgb.shape[2] * rgb.shape[3] % 4 == 0:
            # normalize(rgb / 255) = rgb * (1 / (255 std)) - mean / std: one launch per image instead of three
            st = self.__dict__.get("_norm_affine")      # (constants of the module: computed once per device, not per frame)
            if st is None or st[0].device != rgb.device:
                std, mean = self.normalize.std.detach().double(), self.normalize.mean.detach().double()
                st = ((1.0 / (255.0 * std)).float().reshape(-1).to(rgb.device), (-mean / std).float().reshape(-1).to(rgb.device))
                object.__setattr__(self, "_norm_affine", st)
            s_, t_ = st
            return self.conv_backbone(ops.channel_affine(rgb, s_, t_))
        return self.conv_backbone(self.normalize(rgb / 255.))

    def classify(self, x1, x2):
        """Both attention poolings + Linear(1024, 1) + sigmoid on the fused HIP path (batch 1, eval), else None."""
        if not self.training and x1.is_cuda and x1.shape[0] == 1:
            # both pooled vectors land in one (1, 1024) buffer (no cat), classifier = one small launch (no library GEMM)
            both = torch.empty((1, 2 * x1.shape[1]), dtype=torch.float32, device=x1.device)
            C = x1.shape[1]
            self.attn1(x1, out=both[:, :C]); self.attn2(x2, out=both[:, C:])
            lin = self.classifier[0]
            return ops.linear_act(both, lin.weight, lin.bias, sigmoid=True)[:, 0]
        return None

    def forward(self, rgb1, rgb2, mask=False):
        x1 = self.trunk(rgb1)
        x2 = self.trunk(rgb2)
        if not mask:
            pred = self.classify(x1, x2)
            if pred is not None:
                return pred
        pred = self.classifier(torch.cat([self.attn1(x1), self.attn2(x2)], dim=1))
        if mask:
            return pred[:, 0], F.interpolate(self.seg_head(x1), scale_factor=4), F.interpolate(self.seg_head(x2), scale_factor=4)
        return pred[:, 0]
