#!/usr/bin/env python3
"""Launch plan of every convolution of the frame (host only, no GPU): tile, LDS, split-K, workgroups, rounds."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lav_amd  # noqa: E402
from lav_amd import _lib  # noqa: E402
from lav_amd._lib import Conv  # noqa: E402
from lav_amd.rgb import RGBBrakePredictionModel, RGBSegmentationModel  # noqa: E402

lib = _lib.load()


def conv_shapes(model, shapes):
    out, hooks = [], []
    for m in model.modules():
        if isinstance(m, (torch.nn.Conv2d, torch.nn.ConvTranspose2d)):
            hooks.append(m.register_forward_hook(lambda mod, inp, o: out.append((mod, tuple(inp[0].shape)))))
    model.train()          # torch path of the mixed modules
    with torch.no_grad():
        model(*[torch.zeros(s) for s in shapes])
    for h in hooks:
        h.remove()
    return out


todo = [("erfnet", c, s) for c, s in conv_shapes(RGBSegmentationModel([4, 6, 7, 10]), [(3, 3, 288, 256)])]
todo += [("brake", c, s) for c, s in conv_shapes(RGBBrakePredictionModel([4, 6, 7, 10]).conv_backbone, [(1, 3, 288, 768)])]
todo += [("brake-tel", c, s) for c, s in conv_shapes(RGBBrakePredictionModel([4, 6, 7, 10]).conv_backbone, [(1, 3, 192, 480)])]
todo += [("bev", c, s) for c, s in conv_shapes(lav_amd.ConvBackbone(64), [(1, 64, 320, 320)])]
todo.append(("heads", torch.nn.Conv2d(384, 256, 3, 1, 1), (1, 384, 160, 160)))
todo.append(("heads", torch.nn.ConvTranspose2d(64, 2, 3, 2, 1, 1), (1, 256, 160, 160)))
for nb in (1, 2, 4):
    todo += [(f"resnet B{nb}", c, s) for c, s in conv_shapes(lav_amd.resnet18(num_channels=384), [(nb, 384, 96, 96)])]
info = (C.c_int * 9)()
seen = set()
tot = 0
for name, conv, shp in todo:
    tr = isinstance(conv, torch.nn.ConvTranspose2d)
    cin, cout = conv.in_channels, conv.out_channels
    in_total = shp[1]
    d = Conv(shp[0], in_total, 0, cin, shp[2], shp[3], cout, conv.kernel_size[0], conv.kernel_size[1], conv.stride[0],
             conv.padding[0], conv.padding[1], conv.dilation[0], conv.dilation[1], int(tr), conv.output_padding[0] if tr else 0,
             cout, 0, 0, 0, 0)
    key = (name, shp, cin, cout, conv.kernel_size, conv.stride, conv.dilation, tr)
    if key in seen:
        continue
    seen.add(key)
    assert lib.lav_conv_tile_info(C.byref(d), info) == 0, lib.lav_last_error()
    oh, ow = C.c_int(), C.c_int()
    lib.lav_conv_out_hw(C.byref(d), C.byref(oh), C.byref(ow))
    MP, MCt, rb, Wst, ROWS, lds, ks, tg, cps = list(info)
    s = conv.stride[0]
    QH, QW = (shp[2], shp[3]) if tr else (oh.value, ow.value)
    ncls = s * s if tr else 1
    if MP == -2:  # small-cin vector kernel: info = {-2, tile width, tile rows, workgroups, ...}
        print(f"{name:10s} {'T' if tr else 'C'} {cin:4d}->{cout:4d} k{conv.kernel_size[0]}x{conv.kernel_size[1]} s{s} d{conv.dilation[0]},{conv.dilation[1]} "
              f"in {shp[0]}x{shp[2]}x{shp[3]:<4d} small-cin (packed fp32 FMA) tile {MCt}x{rb} wgs={Wst:5d} rounds={-(-Wst // 256)}")
        continue
    if MP == -1:  # split-operand kernel (bf16x6): info = {-1, MP, MC, pixel waves, tile width (0 = linearised), LDS, split-K, tap group, tile rows}
        mp, mc, wpx, tw, th = MCt, rb, Wst, ROWS, cps
        pixw = wpx * mp * 32
        nblk = (4 // wpx) * mc
        tiles = (-(-QW // tw)) * (-(-QH // th)) if tw else -(-(QH * QW) // pixw)
        nwg = tiles * (-(-cout // (32 * nblk))) * shp[0] * ncls * max(ks, 1)
        sk = f" stream-K: {nwg} tiles over {-ks} persistent workgroups" if ks < 0 else ""
        print(f"{name:10s} {'T' if tr else 'C'} {cin:4d}->{cout:4d} k{conv.kernel_size[0]}x{conv.kernel_size[1]} s{s} d{conv.dilation[0]},{conv.dilation[1]} "
              f"in {shp[0]}x{shp[2]}x{shp[3]:<4d} split {mp}x{mc}/w{wpx} tile {tw}x{th} G={tg % 100}{' tap pairs' if 100 <= tg < 200 else ''}{' f16x3' if tg >= 200 else ''} lds={lds // 1024:3d}K ks={max(ks, 1):2d} wgs={nwg:5d} rounds={-(-nwg // 256)}{sk} "
              f"pix-util={QH * QW / max(tiles * pixw, 1):.2f}")
        continue
    if MP == 0:   # direct kernel: info = {0, waves, cout blocks per workgroup, -, -, LDS, split-K, ...}
        waves, mcb = MCt, rb
        nwg = -(-(shp[0] * QH * QW) // 32) * (-(-cout // (32 * mcb))) * ncls * ks
        print(f"{name:10s} {'T' if tr else 'C'} {cin:4d}->{cout:4d} k{conv.kernel_size[0]}x{conv.kernel_size[1]} s{s} d{conv.dilation[0]},{conv.dilation[1]} "
              f"in {shp[0]}x{shp[2]}x{shp[3]:<4d} direct waves={waves:2d} cout-blocks={mcb} lds={lds // 1024:3d}K ks={ks:2d} wgs={nwg:5d} rounds={-(-nwg // 256)}")
        continue
    PIXW = 128 * MP
    xt = QH * ((QW + PIXW - 1) // PIXW) if rb else (QH * QW + PIXW - 1) // PIXW
    nwg = xt * ((cout + 32 * MCt - 1) // (32 * MCt)) * shp[0] * ncls * ks
    util = QH * QW / (xt * PIXW)
    print(f"{name:10s} {'T' if tr else 'C'} {cin:4d}->{cout:4d} k{conv.kernel_size[0]}x{conv.kernel_size[1]} s{s} d{conv.dilation[0]},{conv.dilation[1]} "
          f"in {shp[0]}x{shp[2]}x{shp[3]:<4d} tile {MP}x{MCt} rb={rb} lds={lds // 1024:3d}K ks={ks:2d} tg={tg} cps={cps} wgs={nwg:5d} rounds={-(-nwg // 256)} pix-util={util:.2f}")
