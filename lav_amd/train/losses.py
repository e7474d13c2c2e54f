"""Losses of train_lidar / train_bev (lav/models/loss.py:5-27, lav/lav_final_v2.py:173-220,
lav/lav_privileged_v2.py:131-141), as pure functions of the model outputs."""
from __future__ import annotations

import torch
from torch import nn
from torch.nn import functional as F


class DetLoss(nn.Module):
    """CenterNet-style detection loss: focal-like re-weighted BCE on the heat maps, SmoothL1 on size / orientation
    maps weighted by the per-pixel maximum of the target heat map (loss.py:5-27)."""

    def forward(self, pred_heatmaps, heatmaps, pred_sizemaps, sizemaps, pred_orimaps, orimaps):
        size_w = heatmaps.max(dim=1, keepdim=True)[0]
        p_det = torch.sigmoid(pred_heatmaps * (1 - 2 * heatmaps))
        bce = F.binary_cross_entropy_with_logits(pred_heatmaps, heatmaps, reduction="none")
        det_loss = (bce * p_det).mean() / p_det.mean()
        box_loss = (size_w * F.smooth_l1_loss(pred_sizemaps, sizemaps, reduction="none")).mean() / size_w.mean()
        ori_loss = (size_w * F.smooth_l1_loss(pred_orimaps, orimaps, reduction="none")).mean() / size_w.mean()
        return det_loss, box_loss, ori_loss


def build_seg_mask(w=320, h=320, cx=160, cy=280, radius_x=240, radius_y=240):
    """Gaussian pixel weights of the road-segmentation loss centred on the ego pixel (lav_final_v2.py:261-270)."""
    gx = (-((torch.arange(w)[:, None] - cx) / radius_x) ** 2).exp()
    gy = (-((torch.arange(h)[:, None] - cy) / radius_y) ** 2).exp()
    return (gx[None] * gy[:, None]).max(dim=-1)[0]


def _pick_cmd(locs, cmds):
    """(B, num_cmds, T, 2) -> (B, T, 2): the branch of each sample's command."""
    T = locs.size(2)
    return locs.gather(1, cmds.expand(T, 2, 1, -1).permute(3, 2, 0, 1)).squeeze(1)


def bev_losses(out, ego_locs, cmds, idxs, cfg, branch_weights, other_weight):
    """train_bev's loss terms (lav_privileged_v2.py:131-141).  out = BEVPlanner.forward's tuple."""
    other_next_locs, other_cast_locs, other_cast_cmds, ego_plan_locs, ego_cast_locs, ego_cast_cmds = out
    tgt = ego_locs[:, 1:]
    per = F.l1_loss(ego_plan_locs, tgt[:, None, None].repeat(1, cfg.num_plan_iter, cfg.num_cmds, 1, 1), reduction="none").mean(dim=[1, 2, 3, 4])
    plan_loss = torch.mean(per[idxs] * branch_weights[cmds[idxs]])
    ego_cast_loss = F.l1_loss(_pick_cmd(ego_cast_locs, cmds), tgt, reduction="none").mean(dim=[1, 2]).mean()
    oc = F.l1_loss(other_cast_locs, other_next_locs.unsqueeze(1).repeat(1, cfg.num_cmds, 1, 1), reduction="none").mean(dim=[2, 3])
    other_cast_loss = oc.min(1)[0].mean()
    label = (1. - cfg.cmd_smooth) * F.one_hot(cmds, cfg.num_cmds) + cfg.cmd_smooth / cfg.num_cmds
    cmd_loss = F.binary_cross_entropy(ego_cast_cmds, label)
    loss = plan_loss + ego_cast_loss + other_cast_loss * other_weight + cmd_loss * cfg.cmd_weight
    return loss, dict(plan_loss=plan_loss, ego_cast_loss=ego_cast_loss, other_cast_loss=other_cast_loss, cmd_loss=cmd_loss)


def lidar_losses(det_criterion, lidar_out, uni_out, heatmaps, sizemaps, orimaps, seg_bev, seg_mask, ego_locs, cmds, idxs, cfg,
                 branch_weights):
    """train_lidar's loss terms (lav_final_v2.py:173-220)."""
    _, pred_heatmaps, pred_sizemaps, pred_orimaps, pred_bev = lidar_out
    (other_next_locs, other_cast_locs, other_cast_cmds, other_cast_locs_expert, other_cast_cmds_expert,
     _ego_next, ego_plan_locs, ego_cast_locs, ego_cast_cmds, ego_cast_locs_expert, ego_plan_locs_expert) = uni_out
    hm_loss, box_loss, ori_loss = det_criterion(pred_heatmaps, heatmaps, pred_sizemaps, sizemaps, pred_orimaps, orimaps)
    det_loss = hm_loss + cfg.box_weight * box_loss + cfg.ori_weight * ori_loss
    seg_loss = torch.mean(F.binary_cross_entropy(pred_bev, seg_bev, reduction="none") * seg_mask) * cfg.seg_weight
    expert = _pick_cmd(ego_plan_locs_expert[:, -1], cmds).unsqueeze(1).unsqueeze(1).repeat(1, cfg.num_plan_iter, cfg.num_cmds, 1, 1)
    plan_loss = torch.mean(F.l1_loss(ego_plan_locs, expert, reduction="none").mean(dim=[1, 2, 3, 4]) * branch_weights[cmds])
    if cfg.distill:
        ego_cast_loss = F.l1_loss(ego_cast_locs, ego_cast_locs_expert)
        other_cast_loss = F.l1_loss(other_cast_locs, other_cast_locs_expert)
        cmd_loss = F.binary_cross_entropy(other_cast_cmds, other_cast_cmds_expert)
    else:
        ego_cast_loss = F.l1_loss(_pick_cmd(ego_cast_locs, cmds), ego_locs[:, 1:], reduction="none").mean(dim=[1, 2])[idxs].mean()
        oc = F.l1_loss(other_cast_locs, other_next_locs.unsqueeze(1).repeat(1, cfg.num_cmds, 1, 1), reduction="none").mean(dim=[2, 3])
        other_cast_loss = oc.min(1)[0].mean()
        label = (1. - cfg.cmd_smooth) * F.one_hot(cmds, cfg.num_cmds) + cfg.cmd_smooth / cfg.num_cmds
        cmd_loss = F.binary_cross_entropy(ego_cast_cmds, label)
    mot_loss = plan_loss + ego_cast_loss + other_cast_loss * cfg.other_weight + cmd_loss * cfg.cmd_weight
    if cfg.perceive_only:
        loss = det_loss + seg_loss
    elif cfg.motion_only:
        loss = mot_loss
    else:
        loss = mot_loss + (det_loss + seg_loss) * cfg.perception_weight
    return loss, dict(hm_loss=hm_loss, box_loss=box_loss, ori_loss=ori_loss, seg_loss=seg_loss, plan_loss=plan_loss,
                      ego_cast_loss=ego_cast_loss, other_cast_loss=other_cast_loss, cmd_loss=cmd_loss)
