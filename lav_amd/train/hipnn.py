"""Train-mode layers on liblav_amd's hand-written kernels, as torch.autograd Functions over the C ABI.

bn_act(bn, x, relu_pre/relu_post, residual): nn.BatchNorm2d on batch statistics with the ReLU before it (ConvBackbone's
Conv -> ReLU -> BatchNorm, team_code_v2/models/lidar.py:57-108) or after it and the residual add (ResNet-18's BasicBlock,
lav/models/resnet.py) in lav_bn_train_forward / lav_bn_train_backward: two launches each way instead of the 3-5 torch /
MIOpen launches, 8 passes over the activation instead of 13, bit-reproducible sums.  The module's parameters and running
statistics are the nn.BatchNorm2d's own (state_dict keys unchanged).

On a CPU tensor (the gloo tests, the CPU training baseline) the same function runs the torch modules.
LAV_TRAIN_BN=torch forces the torch path on the GPU too (A/B timing).
"""
from __future__ import annotations

import os

import torch
import torch.nn.functional as F

from .. import _lib
from .. import ops as ops_mod
from ..ops import _ptr, _stream, check

_WS = {}


def _workspace(channels: int, device) -> torch.Tensor:
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    need = _lib.load().lav_bn_train_workspace_bytes(int(channels))
    ws = _WS.get(key)
    if ws is None or ws.numel() < need:
        ws = torch.empty(need, dtype=torch.uint8, device=device)
        _WS[key] = ws
    return ws


def _bn_amax(B, C, hw, device):
    """An ops.Amax for what a BatchNorm launch writes (lav_bn_train_*_amax), or None when the step does not run on fp16 pieces."""
    if train_precision() != _lib.CONV_F16X3 or os.environ.get("LAV_TRAIN_BN_AMAX", "1") == "0":
        return None
    n = _lib.load().lav_bn_train_amax_count(int(B), int(C), int(hw))
    if n <= 0:
        return None
    am = ops_mod.Amax(device, capacity=n, zeroed=False)     # (the launch writes every part)
    am.take(n)
    return am


def _tag(t, am):
    """Attach the bound to the tensor OBJECT it describes, with the tensor's version: a later in-place change (autograd accumulating a
    second gradient into the buffer) voids it (_trusted)."""
    if am is not None:
        am.version = t._version
        t._lav_amax = am
    return t


def _trusted(t):
    am = getattr(t, "_lav_amax", None)
    return am if am is not None and getattr(am, "version", None) == t._version else None


def carry(out, src):
    """`out` holds a subset / a maximum of the values of `src` (a max-pooling): src's bound holds for it."""
    am = _trusted(src)
    if am is not None:
        import copy
        _tag(out, copy.copy(am))      # (shares the parts; its own version stamp)
    return out


class _BnAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, residual, eps, relu_pre, relu_post):
        x = x.contiguous()
        B, C, H, W = x.shape
        res = None if residual is None else residual.contiguous()
        y = torch.empty_like(x)
        save = torch.empty((3, C), dtype=torch.float32, device=x.device)   # mean, biased var, rstd
        ws = _workspace(C, x.device)
        am = _bn_amax(B, C, H * W, x.device)
        check(_lib.load().lav_bn_train_forward_amax(_ptr(x), _ptr(res), _ptr(y), B, C, H * W, _ptr(gamma.contiguous()), _ptr(beta.contiguous()),
                                                    float(eps), int(relu_pre), int(relu_post), _ptr(save[0]), _ptr(save[1]), _ptr(save[2]),
                                                    _ptr(am.buf) if am is not None else None, _ptr(ws), ws.numel(), _stream()), "lav_bn_train_forward")
        _tag(y, am)        # the convolution that reads y takes its scale from here (conv2d below)
        ops_mod.train_work["bn_train_fwd_bytes"] += 4 * x.numel() * (3 + (res is not None))     # x twice, y once (+ the residual)
        ctx.save_for_backward(x, y if relu_post else None, gamma, save)
        ctx.cfg = (bool(relu_pre), bool(relu_post), residual is not None)
        mean, var = save[0], save[1]
        ctx.mark_non_differentiable(mean, var)
        return y, mean, var

    @staticmethod
    def backward(ctx, dy, _dmean, _dvar):
        x, y, gamma, save = ctx.saved_tensors
        relu_pre, relu_post, has_res = ctx.cfg
        dy = dy.contiguous()
        B, C, H, W = x.shape
        dx = torch.empty_like(x)
        # the residual branch's gradient is the masked dy; without relu_post that is dy itself (no extra pass)
        dres = torch.empty_like(x) if (has_res and relu_post) else None
        dgb = torch.empty((2, C), dtype=torch.float32, device=x.device)
        ws = _workspace(C, x.device)
        am = _bn_amax(B, C, H * W, x.device)
        check(_lib.load().lav_bn_train_backward_amax(_ptr(x), _ptr(y), _ptr(dy), B, C, H * W, _ptr(gamma.contiguous()), _ptr(save[0]), _ptr(save[2]),
                                                     int(relu_pre), int(relu_post), _ptr(dx), _ptr(dres), _ptr(dgb[0]), _ptr(dgb[1]),
                                                     _ptr(am.buf) if am is not None else None, _ptr(ws), ws.numel(), _stream()), "lav_bn_train_backward")
        _tag(dx, am)       # the convolution whose output this BatchNorm read takes its gradient's scale from here (_Conv2d.backward)
        # x and dy twice, dx once; relu_post reads y (twice without a residual, else once + the masked gradient written and re-read)
        ops_mod.train_work["bn_train_bwd_bytes"] += 4 * x.numel() * (5 + (2 if relu_post else 0))
        return dx, dgb[0], dgb[1], (dres if dres is not None else dy) if has_res else None, None, None, None


def _use_hip(bn, x) -> bool:
    return (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and bn.training and bn.affine and bn.track_running_stats
            and os.environ.get("LAV_TRAIN_BN", "hip") != "torch")


def bn_act(bn: torch.nn.BatchNorm2d, x: torch.Tensor, relu_pre: bool = False, relu_post: bool = False, residual=None) -> torch.Tensor:
    """relu_post(bn(relu_pre(x)) + residual) with `bn` in train mode (batch statistics, running statistics updated like
    nn.BatchNorm2d: momentum, unbiased variance, num_batches_tracked)."""
    if not _use_hip(bn, x):
        t = F.relu(x) if relu_pre else x
        y = bn(t)
        if residual is not None:
            y = y + residual
        return F.relu(y) if relu_post else y
    y, mean, var = _BnAct.apply(x, bn.weight, bn.bias, residual, bn.eps, relu_pre, relu_post)
    _update_running(bn, mean, var, x.shape[0] * x.shape[2] * x.shape[3])
    return y


def _update_running(bn, mean, var, n):
    with torch.no_grad():
        bn.num_batches_tracked += 1
        m = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked)
        bn.running_mean.mul_(1.0 - m).add_(mean, alpha=m)
        bn.running_var.mul_(1.0 - m).add_(var, alpha=m * n / max(n - 1, 1))


def bn_act_many(bns, x: torch.Tensor, relu_pre: bool = False, relu_post: bool = False) -> torch.Tensor:
    """Several BatchNorm2d modules over consecutive channel blocks of one tensor (BatchNorm is per channel: one launch pair with
    the concatenated parameters is the modules' own normalisations)."""
    if not all(_use_hip(bn, x) for bn in bns) or len({bn.eps for bn in bns}) != 1:
        outs, off = [], 0
        for bn in bns:
            outs.append(bn_act(bn, x[:, off:off + bn.num_features], relu_pre, relu_post))
            off += bn.num_features
        return torch.cat(outs, dim=1)
    y, mean, var = _BnAct.apply(x, torch.cat([bn.weight for bn in bns]), torch.cat([bn.bias for bn in bns]), None, bns[0].eps, relu_pre, relu_post)
    n, off = x.shape[0] * x.shape[2] * x.shape[3], 0
    for bn in bns:
        _update_running(bn, mean[off:off + bn.num_features], var[off:off + bn.num_features], n)
        off += bn.num_features
    return y


def conv_relu_bn(seq, x):
    """A run of [Conv, ReLU, BatchNorm2d] triples (ConvBackbone's stages and up-convolutions) in train mode."""
    mods = list(seq)
    if len(mods) % 3:
        raise RuntimeError("expected [conv, relu, bn] triples")
    for j in range(0, len(mods), 3):
        c = mods[j]
        y = conv_module(c, x) if isinstance(c, torch.nn.Conv2d) else (conv_transpose_module(c, x) if isinstance(c, torch.nn.ConvTranspose2d) else c(x))
        x = bn_act(mods[j + 2], y, relu_pre=True)
    return x


# ------------------------------------------------------------------------------------------------------------------ convolutions
# Round 5: the training graph's 3x3 / 7x7 convolutions on liblav_amd's own kernels.
#   forward        lav_conv2d (split-operand bf16x6 / fp32 plans, the inference kernels) over the LIVE parameter: the packed weights are
#                  re-gathered on the device before the launch (lav_conv_repack through the layer's index map: one small launch)
#   data gradient  the same kernel on the adjoint problem - a transposed convolution with the same weight tensor (strides 1 and 2,
#                  profiles/r05_dgrad_probe.txt)
#   weight gradient lav_conv_wgrad (bf16x6 matrix-core kernels, csrc/conv_wgrad.hip) for the 3x3 layers of stride 1 / 2 and the 7x7
#                  stride-2 stems on maps from 40 x 40 - the BEV backbone, the fused heads convolution, the crops' stems: where the
#                  step spends its convolution time -, torch / MIOpen for the rest (LAV_TRAIN_WGRAD=torch: everywhere)
#   transposed     nn.ConvTranspose2d (the backbone's up-convolutions): torch / MIOpen by default; LAV_TRAIN_CONVT=hip: forward on
#                  lav_conv2d's transposed plan, data gradient = the ordinary convolution with the same weight tensor on lav_conv2d,
#                  weight gradient torch (correct, 1 ms per step slower: not the default)
# LAV_TRAIN_CONV=torch routes everything back to torch.nn.functional (A/B timing, the CPU tests).
_CONV_ENGINES = {}
_CONV_ENGINES_MAX = 256


_train_precision_stack = []


def train_precision() -> int:
    """lav_conv.precision of the training graph's forward / data-gradient / weight-gradient convolutions: the environment's
    LAV_TRAIN_PRECISION if set, else the innermost `use_precision(...)` (the trainers: LAV.train_lidar asks for f16x3, LAV.train_bev for
    bf16x6), else bf16x6.
      f16x3 (round 6)  every split-kernel layer and the weight-gradient kernels on two fp16 pieces per operand and three products -
                       the batch-32 layers are matrix / power bound like the frame's head convolution -, packed weights re-gathered AND
                       re-scaled on the device per step (lav_conv_repack_scratch), every activation and gradient tensor measured once
                       per step (lav_absmax_parts) for the kernels that read it: train_full 132 -> 118 ms per step
      bf16x6 (round 5) three bf16 pieces, six products."""
    name = os.environ.get("LAV_TRAIN_PRECISION") or (_train_precision_stack[-1] if _train_precision_stack else "bf16x6")
    return {"f16x3": _lib.CONV_F16X3, "bf16x6": _lib.CONV_BF16X6}.get(name, 0)


class use_precision:
    """with use_precision("f16x3"): ... around a training step (forward AND backward)."""

    def __init__(self, name: str):
        self.name = name

    def __enter__(self):
        _train_precision_stack.append(self.name)
        return self

    def __exit__(self, *exc):
        _train_precision_stack.pop()
        return False


def _conv_engine(kind, w, stride, padding, dilation, transposed, output_padding):
    """One ConvLayer per (role, geometry, device, stream): the packed buffer is overwritten by every refresh, which is safe because
    pack and launch are stream-ordered and nothing keeps the packed weights beyond its launch."""
    from ..ops import ConvLayer
    dev = w.device
    key = (kind, tuple(w.shape), int(stride), tuple(padding), tuple(dilation), bool(transposed), int(output_padding), dev,
           torch.cuda.current_stream(dev).cuda_stream)
    prec = train_precision()
    key = key + (prec,)
    eng = _CONV_ENGINES.get(key)
    if eng is None:
        if len(_CONV_ENGINES) >= _CONV_ENGINES_MAX:      # (a trainer has ~60 distinct keys; a sweep over shapes must not pin HBM forever)
            _CONV_ENGINES.pop(next(iter(_CONV_ENGINES)))
        eng = ConvLayer(w.detach(), stride=stride, padding=tuple(padding), dilation=tuple(dilation), transposed=transposed,
                        output_padding=output_padding, precision=prec, device=dev)
        _CONV_ENGINES[key] = eng
    eng._src["weight"] = w.detach()
    eng.refresh()
    return eng


def _measure(t: torch.Tensor):
    """ops.Amax holding the maxima of the finite |t| (one launch, 512 parts): LAV_CONV_F16X3's activation scale, measured ONCE per
    tensor and step and handed to every kernel that reads the tensor (forward convolution + weight gradient for x, data gradient +
    weight gradient for dy) instead of once per launch."""
    ops_mod.train_work["absmax_launches"] = ops_mod.train_work.get("absmax_launches", 0) + 1
    am = ops_mod.Amax(t.device, capacity=512, zeroed=False)      # (the launch writes all 512 parts: no fill kernel per measurement)
    check(_lib.load().lav_absmax_parts(_ptr(t), t.numel(), _ptr(am.take(512)), _stream()), "lav_absmax_parts")
    return am


class _Conv2d(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, stride, padding, dilation, am_in=None):
        x = x.contiguous()
        eng = _conv_engine("fwd", w, stride, padding, dilation, False, 0)
        am_x = None
        if train_precision() == _lib.CONV_F16X3 and eng.uses_amax(*x.shape[:1], *x.shape[2:]):
            am_x = am_in if am_in is not None else _measure(x)
        y = eng(x, amax_in=am_x)
        ctx.save_for_backward(x, w)
        ctx.am_x = am_x
        ctx.cfg = (stride, padding, dilation)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        stride, padding, dilation = ctx.cfg
        am_in = _trusted(dy)
        dy = dy.contiguous()
        dx = dw = None
        am_dy = None
        if train_precision() == _lib.CONV_F16X3 and dy.numel() > 0:
            am_dy = am_in if am_in is not None else _measure(dy)
        lav_dgrad = os.environ.get("LAV_TRAIN_DGRAD", "hip") != "torch" and (stride == 1 or os.environ.get("LAV_TRAIN_DGRAD_STRIDED", "hip") == "hip")
        if ctx.needs_input_grad[0] and lav_dgrad:
            kh, kw = w.shape[2], w.shape[3]
            # the adjoint's output_padding = the rows / columns of x the strided forward never reached, PER DIMENSION; the transposed
            # plan takes one value for both and requires it below the stride - anything else goes to torch
            oph = x.shape[2] - ((dy.shape[2] - 1) * stride - 2 * padding[0] + dilation[0] * (kh - 1) + 1)
            opw = x.shape[3] - ((dy.shape[3] - 1) * stride - 2 * padding[1] + dilation[1] * (kw - 1) + 1)
            if oph == opw and 0 <= oph < max(stride, 1) and dy.numel() > 0:
                dx = _conv_engine("dgrad", w, stride, padding, dilation, True, oph)(dy, amax_in=am_dy)
                if dx.shape != x.shape:
                    raise RuntimeError(f"convolution data gradient {tuple(dx.shape)} != input {tuple(x.shape)}")
        need_dx_torch = ctx.needs_input_grad[0] and dx is None
        dw_hip = None
        if ctx.needs_input_grad[1]:
            dw_hip = _wgrad_hip(x, dy, w, stride, padding, dilation, ctx.am_x if ctx.am_x is not None and am_dy is not None else None, am_dy)
        if need_dx_torch or (ctx.needs_input_grad[1] and dw_hip is None):
            gi, gw, _ = torch.ops.aten.convolution_backward(dy, x, w, None, [stride, stride], list(padding), list(dilation), False, [0, 0], 1,
                                                            [need_dx_torch, ctx.needs_input_grad[1] and dw_hip is None, False])
            if need_dx_torch:
                dx = gi
            if dw_hip is None:
                dw = gw
        if dw_hip is not None:
            dw = dw_hip
        return dx, dw, None, None, None, None


def _wgrad_hip(x, dy, w, stride, padding, dilation, am_x=None, am_dy=None):
    """Weight gradient on lav_conv_wgrad where the kernel applies, else None.  am_x, am_dy (ops.Amax of x and dy, both or neither):
    the fp16 two-piece kernels (lav_conv_wgrad_amax)."""
    lib = _lib.load()
    if os.environ.get("LAV_TRAIN_WGRAD", "hip") == "torch" or not hasattr(lib, "lav_conv_wgrad"):
        return None
    cout, cin, kh, kw = w.shape
    B, _, H, W = x.shape
    # (maps of at least 40 x 40 pixels: below that a task has too few 16-pixel steps behind its prologue and MIOpen wins -
    # profiles/r05_wgrad_probe.txt: 64 -> 64 @24x24 x 96 crops 92 vs 68 us)
    shape_ok = kh == kw and ((kh == 3 and stride in (1, 2)) or (kh == 7 and stride == 2)) and tuple(padding) == (kh // 2, kh // 2)
    min_pix = int(os.environ.get("LAV_TRAIN_WGRAD_MINPIX", "1600"))
    if not (shape_ok and tuple(dilation) == (1, 1) and W % (4 * stride) == 0 and H % stride == 0 and dy.shape[2] * dy.shape[3] >= min_pix
            and cin % 64 == 0 and cout % 64 == 0 and dy.shape[2] == H // stride and dy.shape[3] == W // stride):
        return None
    if (stride == 2 and kh == 3 and os.environ.get("LAV_TRAIN_WGRAD_STRIDED", "hip") == "torch") or (
            kh == 7 and os.environ.get("LAV_TRAIN_WGRAD_STEM", "hip") == "torch"):
        return None
    if w.dtype != torch.float32 or not w.is_contiguous() or x.data_ptr() % 16 or dy.data_ptr() % 16:
        return None     # (the kernel writes the dense [cout][cin][k][k] layout and loads 16-byte vectors)
    dw = torch.empty(w.shape, dtype=torch.float32, device=w.device)
    nbytes = lib.lav_conv_wgrad_workspace_bytes(B, cin, cout, H, W, kh, stride)
    ws = ops_mod._workspace("conv_wgrad", nbytes, x.device)
    if am_x is not None and am_dy is not None and os.environ.get("LAV_TRAIN_WGRAD_F16", "1") != "0":
        check(lib.lav_conv_wgrad_amax(_ptr(x), _ptr(dy), B, cin, cout, H, W, kh, stride, _ptr(dw), _ptr(ws), ws.numel(),
                                      _ptr(am_x.buf), am_x.count, _ptr(am_dy.buf), am_dy.count, _stream()), "lav_conv_wgrad_amax")
    else:
        check(lib.lav_conv_wgrad(_ptr(x), _ptr(dy), B, cin, cout, H, W, kh, stride, _ptr(dw), _ptr(ws), ws.numel(), _stream()), "lav_conv_wgrad")
    ops_mod.train_work["conv_wgrad_flops"] = ops_mod.train_work.get("conv_wgrad_flops", 0) + 2 * B * dy.shape[2] * dy.shape[3] * cin * cout * kh * kw
    return dw


def conv2d(x: torch.Tensor, w: torch.Tensor, stride: int = 1, padding=(0, 0), dilation=(1, 1)) -> torch.Tensor:
    """F.conv2d(x, w, None, stride, padding, dilation) with forward / data gradient / weight gradient on liblav_amd where its
    kernels apply (kernel >= 3, float32, HBM), torch otherwise."""
    if isinstance(padding, int):
        padding = (padding, padding)
    if isinstance(dilation, int):
        dilation = (dilation, dilation)
    if (not x.is_cuda or x.dtype != torch.float32 or w.dtype != torch.float32 or w.shape[2] < 3 or not torch.is_grad_enabled()
            or x.numel() == 0 or os.environ.get("LAV_TRAIN_CONV", "hip") == "torch"):
        return F.conv2d(x, w, None, stride, tuple(padding), tuple(dilation))
    # (the bound a producer left on x - a BatchNorm of this module, a max-pool of one - is read HERE, from the object the caller holds)
    return _Conv2d.apply(x, w, int(stride), tuple(padding), tuple(dilation), _trusted(x))


class _ConvT2d(torch.autograd.Function):
    """F.conv_transpose2d(x, w, None, stride, padding, output_padding) - w [cin][cout][k][k] as nn.ConvTranspose2d holds it."""
    @staticmethod
    def forward(ctx, x, w, stride, padding, output_padding):
        x = x.contiguous()
        y = _conv_engine("fwdT", w, stride, padding, (1, 1), True, output_padding)(x)
        ctx.save_for_backward(x, w)
        ctx.cfg = (stride, padding, output_padding)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        stride, padding, output_padding = ctx.cfg
        dy = dy.contiguous()
        dx = dw = None
        if ctx.needs_input_grad[0]:
            # the adjoint of a transposed convolution is the ordinary convolution with the same tensor read as [out = cin][in = cout][k][k]
            dx = _conv_engine("dgradT", w, stride, padding, (1, 1), False, 0)(dy)
            if dx.shape != x.shape:   # (an output_padding larger than the stride's slack cannot occur for nn.ConvTranspose2d)
                raise RuntimeError(f"transposed-convolution data gradient {tuple(dx.shape)} != input {tuple(x.shape)}")
        if ctx.needs_input_grad[1]:
            _, dw, _ = torch.ops.aten.convolution_backward(dy, x, w, None, [stride, stride], list(padding), [1, 1], True, [output_padding, output_padding], 1,
                                                           [False, True, False])
        return dx, dw, None, None, None


def conv_transpose_module(conv: torch.nn.ConvTranspose2d, x: torch.Tensor) -> torch.Tensor:
    """nn.ConvTranspose2d forward in train mode on liblav_amd (kernel >= 2, no bias / groups / dilation; the rest: the module itself).
    OPT-IN (LAV_TRAIN_CONVT=hip): measured 1 ms per train_full step SLOWER than MIOpen's Winograd solvers on the backbone's three
    up-convolutions (129.3 / 128.2 vs 130.0 / 129.5 ms, interleaved) - the default stays torch; tests/test_gpu_train.py holds it to torch."""
    if (conv.bias is not None or conv.groups != 1 or conv.stride[0] != conv.stride[1] or tuple(conv.dilation) != (1, 1) or conv.kernel_size[0] < 2
            or conv.output_padding[0] != conv.output_padding[1] or not x.is_cuda or x.dtype != torch.float32 or not torch.is_grad_enabled()
            or os.environ.get("LAV_TRAIN_CONV", "hip") == "torch" or os.environ.get("LAV_TRAIN_CONVT", "torch") != "hip"):
        return conv(x)
    return _ConvT2d.apply(x, conv.weight, int(conv.stride[0]), tuple(conv.padding), int(conv.output_padding[0]))


def conv_module(conv: torch.nn.Conv2d, x: torch.Tensor) -> torch.Tensor:
    """nn.Conv2d forward in train mode through conv2d() above (modules with a bias, groups or anisotropic strides: the module itself)."""
    if (conv.bias is not None or conv.groups != 1 or conv.stride[0] != conv.stride[1] or conv.padding_mode != "zeros"
            or not isinstance(conv.padding, tuple)):
        return conv(x)
    return conv2d(x, conv.weight, conv.stride[0], conv.padding, conv.dilation)
