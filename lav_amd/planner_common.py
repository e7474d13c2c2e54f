"""Pieces shared by UniPlanner and BEVPlanner: the GRU decoders on liblav_amd and the rotated crop."""
from __future__ import annotations

import torch
import torch.nn.functional as F

from . import ops


def _hip_train(what: str) -> bool:
    """Diagnosis knob (tools/curve_bisect.py): LAV_TRAIN_<what>=torch runs that piece of the TRAINING graph on torch's own ops
    (nn.GRU = MIOpen's RNN, affine_grid + grid_sample) instead of liblav_amd's autograd functions - to tell a kernel's effect on a
    training run apart from everything else.  Inference never consults it."""
    import os
    return os.environ.get("LAV_TRAIN_" + what, "hip") != "torch"


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


def filter_cars(ego_locs, locs, typs):
    """Vehicles ahead of the ego vehicle only (bev_planner_v2.py:279-283; -y is forward)."""
    return typs & ((locs[:, :, 0] - ego_locs[:, 0:1])[..., 1] < 0)


def random_sample(binaries, size):
    """Keep at most `size` set entries per row, chosen uniformly (bev_planner_v2.py:286-299; same RNG calls)."""
    cut = torch.zeros_like(binaries)
    counts = binaries.sum(1).tolist()      # (ONE device->host copy for the rows' counts; the reference pays a sync per row)
    if max(counts, default=0) <= size:
        return binaries.clone()
    for i in range(binaries.size(0)):
        if counts[i] <= size:
            cut[i] = binaries[i]
        else:
            nz = torch.nonzero(binaries[i]).squeeze(1)
            nz = nz[torch.multinomial(torch.ones_like(nz).float(), size)]
            cut[i, nz] = binaries[i, nz]
    return cut


def sample_others(self, ego_locs, locs, oris, typs):
    """Shared front half of BEVPlanner.forward / UniPlanner.forward (bev_planner_v2.py:74-101, uniplanner.py:58-86):
    pick the vehicles to train on and draw their crop jitter.  Returns None when no vehicle qualifies."""
    ego_oris = oris[:, :1]
    locs, oris = locs[:, 1:], oris[:, 1:]
    typs = filter_cars(ego_locs, locs, typs[:, 1:] == 1)          # 1 = vehicle
    if int(typs.float().sum()) == 0:
        return None, locs.size(1)
    typs = random_sample(typs, size=self.max_num_cars)
    flat_locs = (locs[:, :, 1:] - locs[:, :, :1])[typs]
    rel_loc0 = (locs[:, :, 0] - ego_locs[:, None, 0])[typs]
    rel_ori0 = (oris - ego_oris)[typs]
    K = flat_locs.size(0)
    locs_jitter = (torch.rand((K, 2)) * 2 - 1).float().to(locs.device) * self.feature_x_jitter
    locs_jitter[:, 1] = 0
    oris_jitter = (torch.rand((K,)) * 2 - 1).float().to(oris.device) * self.feature_angle_jitter
    other_locs = transform_points(flat_locs - locs_jitter[:, None], -rel_ori0 - oris_jitter)
    sample = torch.nonzero(typs)[:, 0].int()      # which sample's maps each picked vehicle is cropped from
    return dict(typs=typs, sample=sample, crop_locs=rel_loc0 + locs_jitter, crop_oris=rel_ori0 + oris_jitter, other_locs=other_locs), locs.size(1)


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
        if self.training:
            return self._cast_torch(embd)
        w = self._dec(embd.device)["cast"]
        return ops.gru_cast(embd, w["w_ih"], w["w_hh"], w["b_ih"], w["b_hh"], w["mlp_w"], w["mlp_b"], self.num_plan)

    def plan(self, embd, nxp, cast_locs=None, pixels_per_meter=4, crop_size=96, cmd: int = -1, impl: str = "auto"):
        """(B,512),(B,2),(B,num_cmds,T,2) -> (B, num_plan_iter, num_cmds, T, 2)  (uniplanner.py:255-286).
        cmd >= 0 evaluates only that command branch and returns (B, iters, 1, T, 2).  impl: see ops.gru_plan."""
        if cast_locs is None:
            cast_locs = self.cast(embd)
        if self.training:
            return self._plan_torch(embd, nxp, cast_locs.detach(), pixels_per_meter, crop_size, cmd)
        w = self._dec(embd.device)["plan"]
        return ops.gru_plan(embd, nxp, cast_locs.detach(), w["w_ih"], w["w_hh"], w["b_ih"], w["b_hh"], w["mlp_w"],
                            w["mlp_b"], self.num_plan_iter, cmd, pixels_per_meter, crop_size, impl=impl)

    # ---- train mode: the same decoders as differentiable torch ops (uniplanner.py:255-308) ----------------------
    def _cast_torch(self, embd):
        """Six independent GRU(512 -> 64) decoders on the same input == ONE GRU(512 -> 384) whose recurrent matrix is
        block diagonal (gate-major stacking [r | z | n], each 6 x 64 rows).  The combined weights are assembled from the
        six modules' parameters with differentiable torch ops on every call, so gradients reach the original tensors;
        MIOpen then runs one RNN call instead of six (each ~2 ms forward + backward at these sizes)."""
        grus, mlps = self._cast_modules()
        nc, H = len(grus), grus[0].hidden_size
        gate = lambda name, g: torch.cat([getattr(m, name).view(3, H, -1)[g] for m in grus], dim=0)
        w_ih = torch.cat([gate("weight_ih_l0", g) for g in range(3)], dim=0)                                   # (3*nc*H, 512)
        w_hh = torch.cat([torch.block_diag(*[m.weight_hh_l0.view(3, H, H)[g] for m in grus]) for g in range(3)], dim=0)
        b_ih = torch.cat([torch.cat([m.bias_ih_l0.view(3, H)[g] for m in grus]) for g in range(3)])
        b_hh = torch.cat([torch.cat([m.bias_hh_l0.view(3, H)[g] for m in grus]) for g in range(3)])
        B = embd.size(0)
        u = embd[:, None].expand(-1, self.num_plan, -1).contiguous()
        h0 = embd.new_zeros((1, B, nc * H))
        if embd.is_cuda and _hip_train("GRU"):   # liblav_amd's sequence GRU: the input is the same at every step, so it is projected once
            out = ops.gru_seq(F.linear(embd, w_ih, b_ih), h0[0], w_hh, b_hh, self.num_plan)                       # (B, T, nc*H)
        else:
            out, _ = torch._VF.gru(u, h0, [w_ih, w_hh, b_ih, b_hh], True, 1, 0.0, self.training, False, True)
        out = out.view(B, self.num_plan, nc, H)
        w = torch.stack([m.weight for m in mlps])                                                               # (nc, 2, H)
        b = torch.stack([m.bias for m in mlps])                                                                 # (nc, 2)
        step = torch.einsum("btch,cdh->bctd", out, w) + b[None, :, None, :]
        return torch.cumsum(step, dim=2)

    def _plan_torch(self, embd, nxp, plan_loc, pixels_per_meter, crop_size, cmd=-1):
        """The reference loops over the six command branches (uniplanner.py:264-275); they share the GRU and never interact,
        so they are run as ONE batched GRU call per refinement iteration (sequences ordered branch-major) - MIOpen's RNN
        costs ~2 ms per call forward + backward whatever the batch, which made 30 calls the bulk of a training step."""
        B, T = embd.size(0), self.num_plan
        if cmd >= 0:
            plan_loc = plan_loc[:, cmd:cmd + 1]
        nb = plan_loc.size(1)
        u0 = (nxp * pixels_per_meter / crop_size * 2 - 1)[:, None].expand(-1, T, -1).repeat(nb, 1, 1)        # (nb*B, T, 2)
        h0 = embd.repeat(nb, 1)[None].contiguous()                                                            # (1, nb*B, 512)
        outs = []
        for _ in range(self.num_plan_iter):
            u = torch.cat([u0, plan_loc.transpose(0, 1).reshape(nb * B, T, 2)], dim=2)
            if u.is_cuda and _hip_train("GRU"):
                g = self.plan_gru
                hseq = ops.gru_seq(F.linear(u, g.weight_ih_l0, g.bias_ih_l0), h0[0], g.weight_hh_l0, g.bias_hh_l0, T)
            else:
                hseq = self.plan_gru(u, h0)[0]
            step = torch.cumsum(self.plan_mlp(hseq), dim=1)                                                   # (nb*B, T, 2)
            plan_loc = step.view(nb, B, T, 2).transpose(0, 1) + plan_loc
            outs.append(plan_loc)
        return torch.stack(outs, dim=1)
