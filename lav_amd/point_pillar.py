"""PointPillarNet / DynamicPointNet with the reference's constructor, state_dict keys and call
signature (lav/models/point_pillar.py:12-116), running on liblav_amd's pillar kernels.

The modules only hold parameters; forward() folds the eval-mode BatchNorm1d into the two Linear
layers once and enqueues lav_pillar_scatter (voxelise -> sort -> PointNet on MFMA -> scatter-max ->
canvas).  There is no torch fallback: tensors must live in HBM.
"""
from __future__ import annotations

import os
from typing import Sequence

import torch
from torch import nn

from . import ops


class DynamicPointNet(nn.Module):
    """Parameter holder for `net` = [Linear, BatchNorm1d, ReLU] x len(num_features) (point_pillar.py:12-26)."""

    def __init__(self, num_input: int = 9, num_features: Sequence[int] = (32, 32)):
        super().__init__()
        blocks = []
        width = num_input
        for nf in num_features:
            blocks += [nn.Linear(width, nf), nn.BatchNorm1d(nf), nn.ReLU(inplace=True)]
            width = nf
        self.net = nn.Sequential(*blocks)
        self._folded = None

    def _invalidate(self):
        self._folded = None

    def _apply(self, fn, *a, **k):
        self._invalidate()
        return super()._apply(fn, *a, **k)

    def train(self, mode: bool = True):
        self._invalidate()
        return super().train(mode)

    def _load_from_state_dict(self, *a, **k):
        self._invalidate()
        return super()._load_from_state_dict(*a, **k)

    def folded(self, device):
        """(w1 [K][C], b1 [C], w2 [C][C], b2 [C]) float32 on `device`, BatchNorm (running stats) folded in float64."""
        if self._folded is not None and self._folded[0].device == device:
            return self._folded
        lins = [m for m in self.net if isinstance(m, nn.Linear)]
        bns = [m for m in self.net if isinstance(m, nn.BatchNorm1d)]
        if len(lins) != 2:
            raise RuntimeError("liblav_amd's pillar kernel is built for a 2-layer PointNet")
        out = []
        for lin, bn in zip(lins, bns):
            s = bn.weight.detach().double() / torch.sqrt(bn.running_var.detach().double() + bn.eps)
            w = lin.weight.detach().double() * s[:, None]                       # [out][in]
            b = (lin.bias.detach().double() - bn.running_mean.detach().double()) * s + bn.bias.detach().double()
            out += [w.t().contiguous().float().to(device), b.float().to(device)]  # kernel wants [in][out]
        self._folded = tuple(out)
        return self._folded


class PointPillarNet(nn.Module):
    def __init__(self, num_input=9, num_features=(32, 32), min_x=-10, max_x=70, min_y=-40, max_y=40,
                 pixels_per_meter=4):
        super().__init__()
        self.point_net = DynamicPointNet(num_input, list(num_features))
        self.nx = (max_x - min_x) * pixels_per_meter
        self.ny = (max_y - min_y) * pixels_per_meter
        self.min_x, self.max_x, self.min_y, self.max_y = min_x, max_x, min_y, max_y
        self.pixels_per_meter = pixels_per_meter
        self.num_input = num_input
        self._grid = ops.make_grid(min_x, max_x, min_y, max_y, pixels_per_meter)
        self._amax = {}

    @staticmethod
    def _pack(lidar_list, num_points):
        """list[Tensor(Ni,D)] | Tensor(B,Nmax,D)  ->  (Tensor(B,Nmax,D), [n_b])  (point_pillar.py:92-98)."""
        if torch.is_tensor(num_points):
            num_points = num_points.tolist()
        num_points = [int(n) for n in num_points]
        if torch.is_tensor(lidar_list):
            pts = lidar_list if lidar_list.dim() == 3 else lidar_list[None]
        elif len(lidar_list) == 1:
            pts = lidar_list[0][None]
        else:
            nmax = max(int(p.shape[0]) for p in lidar_list)
            pts = lidar_list[0].new_zeros((len(lidar_list), nmax, lidar_list[0].shape[1]))
            for b, p in enumerate(lidar_list):
                pts[b, : p.shape[0]] = p
        num_points = [min(n, int(pts.shape[1])) for n in num_points]
        return pts, num_points

    def forward_train(self, lidar_list, num_points):
        """Train-mode forward (BatchNorm1d on batch statistics, autograd): liblav_amd does the index work and the two
        torch_scatter ops - lav_pillar_decorate (grid_locations + unique + scatter_mean + decorate, no_grad in the
        reference too, point_pillar.py:95-112) and lav_scatter_max with its arg-max backward; the two Linear + BatchNorm
        layers and the dense index_put are torch ops on the same stream."""
        pts, n = self._pack(lidar_list, num_points)
        decorated, unique_coords, inverse, _ = ops.pillar_decorate(pts, n, self._grid)
        feat = self.point_net.net(decorated)
        feat_max, _ = ops.scatter_max(feat, inverse, unique_coords.shape[0])
        uc = unique_coords.long()
        canvas = torch.zeros((pts.shape[0], feat_max.shape[1], self.ny, self.nx), dtype=feat_max.dtype, device=feat_max.device)
        canvas[uc[:, 0], :, torch.clamp(self.ny - 1 - uc[:, 1], 0, self.ny - 1), torch.clamp(uc[:, 2], 0, self.nx - 1)] = feat_max
        return canvas

    def forward(self, lidar_list, num_points, return_indices: bool = False):
        if self.training:
            return self.forward_train(lidar_list, num_points)
        pts, n = self._pack(lidar_list, num_points)
        if pts.shape[2] + 5 != self.num_input:
            raise RuntimeError(f"points have {pts.shape[2]} columns, PointNet expects {self.num_input - 5}")
        w1, b1, w2, b2 = self.point_net.folded(pts.device)
        # the canvas leaves with a bound of its values (one float per workgroup of the canvas kernel) for LAV_CONV_F16X3 readers; one
        # fixed buffer per batch size and device: HIP graphs hold its address
        am = None
        if not return_indices and os.environ.get("LAV_PILLAR_AMAX", "1") != "0":
            key = (int(pts.shape[0]), str(pts.device))
            am = self._amax.get(key)
            if am is None:
                am = self._amax[key] = ops.Amax(pts.device, capacity=2048)
        return ops.pillar_scatter(pts, n, self._grid, w1, b1, w2, b2, want_indices=return_indices, amax=am)
