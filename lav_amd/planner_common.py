"""Pieces shared by UniPlanner and BEVPlanner: the GRU decoders on liblav_amd and the rotated crop."""
from __future__ import annotations

import torch
import torch.nn.functional as F

from . import ops


def stack_cast_weights(grus, mlps, device):
    """6 x nn.GRU(512,64) + 6 x nn.Linear(64,2) -> the stacked arrays lav_gru_cast takes."""
    g = lambda n: torch.stack([getattr(m, n).detach() for m in grus]).float().contiguous().to(device)
    return dict(w_ih=g("weight_ih_l0"), w_hh=g("weight_hh_l0"), b_ih=g("bias_ih_l0"), b_hh=g("bias_hh_l0"),
                mlp_w=torch.stack([m.weight.detach() for m in mlps]).float().contiguous().to(device),
                mlp_b=torch.stack([m.bias.detach() for m in mlps]).float().contiguous().to(device))


def plan_weights(gru, mlp, device):
    g = lambda n: getattr(gru, n).detach().float().contiguous().to(device)
    return dict(w_ih=g("weight_ih_l0"), w_hh=g("weight_hh_l0"), b_ih=g("bias_ih_l0"), b_hh=g("bias_hh_l0"),
                mlp_w=mlp.weight.detach().float().contiguous().to(device),
                mlp_b=mlp.bias.detach().float().contiguous().to(device))


def crop_feature(features, rel_locs, rel_oris, pixels_per_meter, crop_size, offset_x, offset_y):
    """Rotated crop of the feature map around each actor (team_code_v2/model_inference.py:204-238 /
    uniplanner.py:310-352): theta = k*R(ori) with the (offset_x, offset_y) pivot, bilinear, zeros outside,
    align_corners=True - one liblav_amd kernel instead of affine_grid + grid_sample.  A feature map that was
    `expand`ed over the batch (stride 0) is passed once and shared by all crops."""
    if features.dim() == 4 and features.shape[0] > 1 and features.stride(0) == 0:
        features = features[:1]
    return ops.crop_rotate(features, rel_locs, rel_oris, pixels_per_meter, crop_size, offset_x, offset_y)


def crop_feature_torch(features, rel_locs, rel_oris, pixels_per_meter, crop_size, offset_x, offset_y):
    """The same crop with torch's affine_grid/grid_sample (kept for training mode, which needs autograd)."""
    B, C, H, W = features.shape
    rel_locs = rel_locs.view(-1, 2) * pixels_per_meter
    rel_locs = torch.stack([rel_locs[:, 0] / (H / 2), rel_locs[:, 1] / (W / 2)], dim=-1)
    cos, sin = torch.cos(rel_oris), torch.sin(rel_oris)
    k = crop_size / H
    rot_x = -k * offset_x * cos + k * offset_y * sin + offset_x
    rot_y = -k * offset_x * sin - k * offset_y * cos + offset_y
    theta = torch.stack([torch.stack([k * cos, k * -sin, rot_x + rel_locs[..., 0]], dim=-1),
                         torch.stack([k * sin, k * cos, rot_y + rel_locs[..., 1]], dim=-1)], dim=-2)
    grid = F.affine_grid(theta, (B, C, crop_size, crop_size), align_corners=True)
    return F.grid_sample(features, grid, align_corners=True)


def transform_points(locs, oris):
    """model_inference.py:240-251: rotate (.., T, 2) waypoints by -ori (row-vector convention)."""
    cos, sin = torch.cos(oris), torch.sin(oris)
    R = torch.stack([torch.stack([cos, sin], dim=-1), torch.stack([-sin, cos], dim=-1)], dim=-2)
    return locs @ R


class DecoderMixin:
    """cast()/plan() on the HIP GRU kernels.  Expects self.num_plan, self.num_plan_iter, self.num_cmds,
    self.plan_gru, self.plan_mlp and a _cast_modules() -> (grus, mlps) hook."""

    def _dec(self, device):
        d = getattr(self, "_dec_cache", None)
        if d is None or d["device"] != device:
            grus, mlps = self._cast_modules()
            d = dict(device=device, cast=stack_cast_weights(grus, mlps, device), plan=plan_weights(self.plan_gru, self.plan_mlp, device))
            object.__setattr__(self, "_dec_cache", d)
        return d

    def _drop_dec(self):
        object.__setattr__(self, "_dec_cache", None)

    def cast(self, embd, mode="ego"):
        """(B,512) -> (B, num_cmds, num_plan, 2).  Both modes use the *_ego GRUs, as the reference does
        (uniplanner.py:296-300)."""
        w = self._dec(embd.device)["cast"]
        return ops.gru_cast(embd, w["w_ih"], w["w_hh"], w["b_ih"], w["b_hh"], w["mlp_w"], w["mlp_b"], self.num_plan)

    def plan(self, embd, nxp, cast_locs=None, pixels_per_meter=4, crop_size=96, cmd: int = -1):
        """(B,512),(B,2),(B,num_cmds,T,2) -> (B, num_plan_iter, num_cmds, T, 2)  (uniplanner.py:255-286).
        cmd >= 0 evaluates only that command branch and returns (B, iters, 1, T, 2)."""
        if cast_locs is None:
            cast_locs = self.cast(embd)
        w = self._dec(embd.device)["plan"]
        return ops.gru_plan(embd, nxp, cast_locs.detach(), w["w_ih"], w["w_hh"], w["b_ih"], w["b_hh"], w["mlp_w"],
                            w["mlp_b"], self.num_plan_iter, cmd, pixels_per_meter, crop_size)
