"""LiDARModel / ConvBackbone / Head with the reference's constructors and state_dict keys
(team_code_v2/models/lidar.py:8-161), evaluated with liblav_amd's MFMA convolution.

Each Conv -> ReLU -> BatchNorm triple of the reference becomes ONE lav_conv2d launch with the ReLU and the
eval-mode BatchNorm affine in its epilogue; the three up-convolutions write straight into their channel
window of the (B,384,160,160) feature map (no torch.cat); the four heads' first convolutions run as a single
384->256 convolution (the 39 MB feature map is read once instead of four times).
"""
from __future__ import annotations

import torch
from torch import nn
import torch.nn.functional as F

from . import _lib
from . import ops
from .ops import Amax, ConvLayer, GroupedDeconv, PointwiseUpconv
from .point_pillar import PointPillarNet

_NORM = dict(eps=1e-3, momentum=0.01)


def _refreshable(obj):
    """Every layer object with a refresh() inside an engine (nested dicts / lists / tuples)."""
    if isinstance(obj, dict):
        for v in obj.values():
            yield from _refreshable(v)
    elif isinstance(obj, (list, tuple)):
        for v in obj:
            yield from _refreshable(v)
    elif hasattr(obj, "refresh"):
        yield obj


class _Engine(nn.Module):
    """Mixin: cache of packed ConvLayers.  When the parameters may have changed IN PLACE (a train()/eval() toggle around optimiser
    steps, load_state_dict) the cached engine is marked stale and re-packed on the device from the live parameters at its next use
    (ConvLayer.refresh: one gather launch per layer - the trainer's per-step log inference, lav_final_v2.py:228-236, used to rebuild
    ~45 layers on the host per step); when the tensors themselves may have been replaced (_apply: .to() / .float()) it is dropped."""

    def _drop(self):
        object.__setattr__(self, "_eng", None)
        object.__setattr__(self, "_stale", False)

    def _mark_stale(self):
        if self.__dict__.get("_eng") is not None:
            object.__setattr__(self, "_stale", True)

    def _tensor_ids(self):
        return tuple(id(t) for t in self.parameters()) + tuple(id(t) for t in self.buffers())

    def _fresh(self, device):
        """The cached engine for `device`, brought up to date; None when there is none (or when a parameter OBJECT was replaced -
        load_state_dict(assign=True), module.weight = nn.Parameter(...) - since the engine was packed: the layers re-pack from
        the tensors they were built from, so such an engine is dropped and rebuilt from the module)."""
        eng = self.__dict__.get("_eng")
        if eng is None or eng.get("device") != device:
            return None
        if self.__dict__.get("_stale"):
            if eng["tensor_ids"] != self._tensor_ids():
                self._drop()
                return None
            for layer in _refreshable(eng):
                layer.refresh()
            object.__setattr__(self, "_stale", False)
        return eng

    def _apply(self, fn, *a, **k):
        self._drop()
        return super()._apply(fn, *a, **k)

    def train(self, mode: bool = True):
        if mode != self.training:     # packed engines only go stale when the mode really changes (or weights do: _apply / load)
            self._mark_stale()
        return super().train(mode)

    def _load_from_state_dict(self, *a, **k):
        self._mark_stale()
        return super()._load_from_state_dict(*a, **k)

    def _need_eval(self):
        if self.training:
            raise NotImplementedError(f"{type(self).__name__}: the fused HIP path is eval-only; train mode runs forward_train")


def _bn_tuple(bn: nn.BatchNorm2d):
    return (bn.running_mean, bn.running_var, bn.weight, bn.bias)


def _stage(cin, cout, n):
    """n x [Conv3x3(no bias) , ReLU, BatchNorm(eps 1e-3)], first conv stride 2 (lidar.py:57-108)."""
    mods = []
    for j in range(n):
        mods += [nn.Conv2d(cin if j == 0 else cout, cout, 3, 2 if j == 0 else 1, 1, bias=False),
                 nn.ReLU(inplace=True), nn.BatchNorm2d(cout, **_NORM)]
    return nn.Sequential(*mods)


def _up(cin, cout, k, s, p=0, op=0):
    return nn.Sequential(nn.ConvTranspose2d(cin, cout, k, s, p, op, bias=False), nn.ReLU(inplace=True),
                         nn.BatchNorm2d(cout, **_NORM))


class ConvBackbone(_Engine):
    def __init__(self, num_feature=64, norm_cfg=None):
        super().__init__()
        f = num_feature
        self.conv1 = _stage(f, f, 4)
        self.conv2 = _stage(f, 2 * f, 6)
        self.conv3 = _stage(2 * f, 2 * f, 6)
        self.upconv1 = _up(f, 2 * f, 1, 1)
        self.upconv2 = _up(2 * f, 2 * f, 4, 2, 1)
        self.upconv3 = _up(2 * f, 2 * f, 4, 4, 1, 2)
        self.out_channels = 6 * f
        self._drop()

    def _engine(self, device):
        """The packed layers for the calling thread's inference precision (ops.infer_precision: the frame pipelines ask for
        LAV_CONV_F16X3, a trainer's log inference gets the default) - one engine per precision, cached side by side."""
        prec = ops.infer_precision()
        eng = self._fresh(device) or dict(device=device)
        key = ("backbone", prec)
        if key in eng:
            return eng[key]
        def stage(seq):
            out = []
            for j in range(0, len(seq), 3):
                conv, bn = seq[j], seq[j + 2]
                out.append(ConvLayer(conv.weight, stride=conv.stride[0], padding=conv.padding, bn=_bn_tuple(bn),
                                     bn_eps=bn.eps, relu_pre=True, precision=prec, device=device))
            return out
        ups = []
        off = 0
        for seq in (self.upconv1, self.upconv2, self.upconv3):
            ct, bn = seq[0], seq[2]
            if prec == _lib.CONV_F16X3 and PointwiseUpconv.takes(ct):
                # (round 6) kernel == stride: one input pixel and tap per output pixel - the 1x1 and the 4x4 / stride-4 layer on the
                # pointwise kernel (exact fp32, lav_upconv_pointwise) instead of an implicit-GEMM plan with a 4-8 step K loop
                ups.append(PointwiseUpconv(ct, bn, relu_pre=True, out_c_total=self.out_channels, out_c_offset=off, device=device))
            else:
                ups.append(ConvLayer(ct.weight, stride=ct.stride[0], padding=ct.padding, transposed=True,
                                     output_padding=ct.output_padding[0], bn=_bn_tuple(bn), bn_eps=bn.eps, relu_pre=True,
                                     out_c_total=self.out_channels, out_c_offset=off, precision=prec, device=device))
            off += ct.weight.shape[1]
        eng[key] = dict(s1=stage(self.conv1), s2=stage(self.conv2), s3=stage(self.conv3), ups=ups, amax={}, f16=prec == _lib.CONV_F16X3)
        eng["tensor_ids"] = self._tensor_ids()   # recorded where the engine is built, not at its first use
        object.__setattr__(self, "_eng", eng)
        return eng[key]

    def forward_train(self, x):
        """Train mode (autograd, BatchNorm on batch statistics): the nn modules themselves (lidar.py:110-143)."""
        from .train.hipnn import conv_relu_bn   # ReLU + batch-statistics BatchNorm as one fused forward / backward pair
        f1 = conv_relu_bn(self.conv1, x)
        f2 = conv_relu_bn(self.conv2, f1)
        f3 = conv_relu_bn(self.conv3, f2)
        return torch.cat([conv_relu_bn(self.upconv1, f1), conv_relu_bn(self.upconv2, f2), conv_relu_bn(self.upconv3, f3)], dim=1)

    def forward(self, x, out=None):
        """`out`: optional preallocated (B, 6*num_feature, H/2, W/2) buffer the three up-convolutions write into."""
        if self.training:
            return self.forward_train(x)
        self._need_eval()
        e = self._engine(x.device)
        if not e["f16"]:
            feats = []
            for st in (e["s1"], e["s2"], e["s3"]):
                for layer in st:
                    x = layer(x)
                feats.append(x)
            oh, ow = e["ups"][0].out_hw(feats[0].shape[2], feats[0].shape[3])
            if out is None:
                out = torch.empty((x.shape[0], self.out_channels, oh, ow), dtype=torch.float32, device=x.device)
            for up, f in zip(e["ups"], feats):
                up(f, out=out)
            return out
        # LAV_CONV_F16X3: every layer leaves the maxima of what it writes for the layers that read it (lav_conv2d_amax) - none of
        # them measures its input; the three up-convolutions leave the feature map's for the heads and the crops' stems
        B = x.shape[0]
        ams = e["amax"].get(tuple(x.shape))
        if ams is None:
            ams = e["amax"][tuple(x.shape)] = [Amax(x.device) for _ in range(len(e["s1"]) + len(e["s2"]) + len(e["s3"]) + 1)]
        stages = (e["s1"], e["s2"], e["s3"])
        feats, feat_am, am_prev, i = [], [], ops.amax_of(x), 0
        for si, st in enumerate(stages):
            for li, layer in enumerate(st):
                oh, ow = layer.out_hw(x.shape[2], x.shape[3])
                readers = ([st[li + 1]] if li + 1 < len(st) else ([stages[si + 1][0]] if si + 1 < len(stages) else [])) + \
                          ([e["ups"][si]] if li + 1 == len(st) else [])
                am = ams[i].reset() if any(r.uses_amax(B, oh, ow) for r in readers) else None
                x = layer(x, amax_in=am_prev, amax_out=am)
                am_prev, i = am, i + 1
            feats.append(x); feat_am.append(am_prev)
        oh, ow = e["ups"][0].out_hw(feats[0].shape[2], feats[0].shape[3])
        if out is None:
            out = torch.empty((B, self.out_channels, oh, ow), dtype=torch.float32, device=x.device)
        am_out = ams[-1].reset()
        for up, f, am in zip(e["ups"], feats, feat_am):
            up(f, out=out, amax_in=am, amax_out=am_out)
        out._lav_amax = am_out
        return out


class Head(_Engine):
    def __init__(self, num_input, num_output, num_hidden=64, norm_cfg=None, output_activation=nn.Identity()):
        super().__init__()
        self.net = nn.Sequential(
            nn.Conv2d(num_input, num_hidden, 3, 1, 1, bias=False),
            nn.ReLU(inplace=True),
            nn.BatchNorm2d(num_hidden, **_NORM),
            nn.ConvTranspose2d(num_hidden, num_output, 3, 2, 1, 1),
        )
        self.output_activation = output_activation
        self._drop()

    @property
    def _sigmoid(self):
        return self.output_activation is torch.sigmoid or isinstance(self.output_activation, nn.Sigmoid)

    def deconv_layer(self, device, in_c_total=None, in_c_offset=0):
        ct = self.net[3]
        if not (self._sigmoid or isinstance(self.output_activation, nn.Identity)):
            raise RuntimeError("Head.output_activation must be identity or sigmoid for the fused epilogue")
        return ConvLayer(ct.weight, stride=2, padding=1, transposed=True, output_padding=1, bias=ct.bias,
                         sigmoid=self._sigmoid, in_c_total=in_c_total, in_c_offset=in_c_offset, device=device)

    def forward(self, x):
        if self.training:
            from .train.hipnn import bn_act
            return self.output_activation(self.net[3](bn_act(self.net[2], self.net[0](x), relu_pre=True)))
        if self._fresh(x.device) is None:
            conv, bn = self.net[0], self.net[2]
            eng = dict(device=x.device,
                       conv=ConvLayer(conv.weight, padding=1, bn=_bn_tuple(bn), bn_eps=bn.eps, relu_pre=True, device=x.device),
                       deconv=self.deconv_layer(x.device))
            eng["tensor_ids"] = self._tensor_ids()   # recorded where the engine is built, not at its first use
            object.__setattr__(self, "_eng", eng)
        return self._eng["deconv"](self._eng["conv"](x))


class LiDARModel(_Engine):
    def __init__(self, num_input=9, num_features=(32, 32), backbone="swin", min_x=-10, max_x=70, min_y=-40, max_y=40,
                 pixels_per_meter=4):
        super().__init__()
        self.point_pillar_net = PointPillarNet(num_input, list(num_features), min_x=min_x, max_x=max_x, min_y=min_y,
                                               max_y=max_y, pixels_per_meter=pixels_per_meter)
        nf = num_features[-1]
        if backbone != "cnn":
            raise NotImplementedError(backbone)
        self.backbone = ConvBackbone(num_feature=nf)
        self.center_head = Head(6 * nf, 2)
        self.box_head = Head(6 * nf, 2)
        self.ori_head = Head(6 * nf, 2)
        self.seg_head = Head(6 * nf, 3, output_activation=torch.sigmoid)
        self._drop()

    def _head_engine(self, names, device):
        """Fused engine of a subset of the heads: ONE convolution 384 -> 64*len(names) (the 39 MB feature map is read once)
        and ONE grouped transposed convolution 64*len -> sum(outputs) (lav_deconv_grouped: every head's tail reads its own
        64 channels); a sigmoid head must come last (lidar.py:30-33,159-161)."""
        prec = ops.infer_precision()
        if prec == 0 and getattr(self, "heads_precision", 0):   # (round 5's per-model knob, still honoured when somebody sets it)
            prec = self.heads_precision
        eng = self._fresh(device) or dict(device=device)
        names = (names, prec)
        if names not in eng:
            hs = [getattr(self, n) for n in names[0]]
            for h in hs:
                if not (h._sigmoid or isinstance(h.output_activation, nn.Identity)):
                    raise RuntimeError("Head.output_activation must be identity or sigmoid for the fused epilogue")
            sig = [h._sigmoid for h in hs]
            if any(sig[:-1]):
                raise RuntimeError("fused head deconvolution expects the sigmoid head last")
            w = lambda: torch.cat([h.net[0].weight.detach() for h in hs], dim=0)     # (callables: re-assembled by ConvLayer.refresh)
            bn = lambda: tuple(torch.cat([getattr(h.net[2], n).detach() for h in hs]) for n in ("running_mean", "running_var", "weight", "bias"))
            cts = [h.net[3] for h in hs]
            outs = [ct.weight.shape[1] for ct in cts]
            eng[names] = dict(outs=outs,
                              conv=ConvLayer(w, padding=1, bn=bn, bn_eps=hs[0].net[2].eps, relu_pre=True, device=device,
                                             precision=prec),   # (the frame pipelines ask for LAV_CONV_F16X3)
                              deconv=GroupedDeconv(cts, sigmoid_from=sum(outs[:-1]) if sig[-1] else -1, device=device))
            eng["tensor_ids"] = self._tensor_ids()   # recorded where the engine is built, not at its first use
            object.__setattr__(self, "_eng", eng)
        return self._eng[names]

    ALL_HEADS = ("center_head", "box_head", "ori_head", "seg_head")

    def heads(self, features, names=ALL_HEADS):
        """Outputs of the named heads (default: center, box, ori, seg) from the shared feature map.  The frame pipeline
        asks for the three detection heads first - the others branch waits on them - and for the segmentation head
        on a side stream."""
        if self.training:
            return self._heads_train(features, names)
        e = self._head_engine(tuple(names), features.device)
        fused = e["deconv"](e["conv"](features, amax_in=ops.amax_of(features)))          # (B, sum(outs), 2H, 2W)
        return tuple(torch.split(fused, e["outs"], dim=1))

    def _heads_train(self, features, names):
        """Train mode: the heads' first convolutions (384 -> 64 each, lidar.py:147-161) as ONE convolution 384 -> 64*len(names)
        over the concatenated weights - the feature map is read once forward and its gradient is produced by one data-gradient
        convolution instead of len(names) of them plus their sum; autograd splits the weight gradient back.  BatchNorm is per
        channel, so normalising the concatenated channels with the concatenated parameters is the heads' own BatchNorms."""
        hs = [getattr(self, n) for n in names]
        if len(hs) == 1 or not features.is_cuda:
            return tuple(h(features) for h in hs)
        from .train.hipnn import bn_act_many, conv2d
        mid = conv2d(features, torch.cat([h.net[0].weight for h in hs], dim=0), 1, 1)
        mid = bn_act_many([h.net[2] for h in hs], mid, relu_pre=True)
        outs, off = [], 0
        for h in hs:
            c = h.net[0].out_channels
            outs.append(h.output_activation(h.net[3](mid[:, off:off + c])))
            off += c
        return tuple(outs)

    def forward(self, lidars, num_points):
        features = self.backbone(self.point_pillar_net(lidars, num_points))
        return (features, *self.heads(features))
