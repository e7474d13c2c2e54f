"""The LAV trainer (lav/lav_final_v2.py:19-270 `train_lidar`, lav/lav_privileged_v2.py:16-160 `train_bev`) with one
process per GPU: when torch.distributed is initialised the trainable modules are wrapped in DistributedDataParallel
(bucketed gradient all-reduce over RCCL/xGMI overlapped with backward; BatchNorm statistics stay per rank and buffers
are broadcast from rank 0, which is what the reference's nn.DataParallel replicas amount to)."""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List

import torch
import torch.distributed as dist
from torch import nn, optim
from torch.optim.lr_scheduler import StepLR

from .. import synth
from ..bev_planner import BEVPlanner
from ..lidar import LiDARModel
from ..uniplanner import UniPlanner
from .losses import DetLoss, bev_losses, build_seg_mask, lidar_losses


@dataclass
class TrainConfig:
    """config_v2.yaml's training keys (defaults = the released file) + the command-line switches of train_*_v2.py."""
    num_plan: int = 20
    num_cmds: int = 6
    seg_channels: List[int] = field(default_factory=lambda: [4, 6, 7, 10])
    crop_size: int = 96
    num_plan_iter: int = 5
    cmd_weight: float = 0.1
    cmd_smooth: float = 0.2
    other_weight: float = 0.5
    backbone: str = "cnn"
    min_x: float = -10
    max_x: float = 70
    min_y: float = -40
    max_y: float = 40
    pixels_per_meter: int = 4
    max_lidar_points: int = 120000
    num_frame_stack: int = 2
    branch_weights: List[float] = field(default_factory=lambda: [5, 5, 5, 1, 1, 1])
    feature_x_jitter: float = 1.5
    feature_angle_jitter: float = 20
    use_others_to_train: bool = True
    box_weight: float = 1.0
    ori_weight: float = 1.0
    seg_weight: float = 2.0
    perception_weight: float = 4.0
    point_painting: bool = True
    num_features: List[int] = field(default_factory=lambda: [64, 64])
    distill: bool = True
    lr: float = 3e-4
    perceive_only: bool = False
    conv_precision: str = ""      # "" = the trainer's default (train_lidar: f16x3, train_bev: bf16x6); LAV_TRAIN_PRECISION overrides
    motion_only: bool = False
    log_inference: bool = True     # the reference runs one eval-mode inference of sample 0 per step, for its logs only
    log_every: int = 100           # (lav_final_v2.py:228-236; logged every --num-per-log = 100 steps): here it runs on those steps
    seed: int = 2021


class _Student(nn.Module):
    """LiDARModel + UniPlanner behind ONE forward, so that DistributedDataParallel sees a single autograd graph."""

    def __init__(self, lidar_model, uniplanner):
        super().__init__()
        self.lidar_model, self.uniplanner = lidar_model, uniplanner

    def forward(self, lidars, num_points, bev, ego_locs, locs, oris, nxps, typs):
        lidar_out = self.lidar_model(lidars, num_points)
        uni_out = self.uniplanner(lidar_out[0], bev, ego_locs, locs, oris, nxps, typs)
        return lidar_out, uni_out


def _scalars(loss, terms):
    """Loss terms as Python floats with ONE device->host copy (the reference pays one float(tensor) sync per term)."""
    keys = list(terms)
    vals = torch.stack([loss.detach()] + [terms[k].detach() for k in keys]).tolist()
    return dict(loss=vals[0], **dict(zip(keys, vals[1:])))


def _ddp(module, device, find_unused=False):
    """find_unused: the student's autograd graph is not the same on every rank and step - a shard without an eligible
    vehicle leaves cast_cmd_pred out of the loss (UniPlanner.forward's `pick is None` branch), --perceive-only leaves the
    whole planner out - and DistributedDataParallel would otherwise wait for those gradients for ever."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return module
    ids = [device.index] if device.type == "cuda" else None
    return nn.parallel.DistributedDataParallel(module, device_ids=ids, bucket_cap_mb=32, gradient_as_bucket_view=True,
                                               find_unused_parameters=find_unused)


class LAV:
    def __init__(self, cfg: TrainConfig, device, what: str = "lidar", checkpoints=None):
        """what: "bev" (train_bev_v2.py) or "lidar" (train_full_v2.py).  checkpoints: optional dict name -> state_dict
        ('bev', 'lidar', 'uniplanner'); missing ones are seeded random weights (the released files are LFS objects)."""
        self.cfg, self.device, self.what = cfg, torch.device(device), what
        self.steps = 0
        ck = checkpoints or {}
        y_off = 1 + cfg.min_x / ((cfg.max_x - cfg.min_x) / 2)
        common = dict(pixels_per_meter=cfg.pixels_per_meter, crop_size=cfg.crop_size, feature_x_jitter=cfg.feature_x_jitter,
                      feature_angle_jitter=cfg.feature_angle_jitter, x_offset=0, y_offset=y_off, num_cmds=cfg.num_cmds,
                      num_plan=cfg.num_plan, num_plan_iter=cfg.num_plan_iter)
        self.bev_planner = BEVPlanner(num_frame_stack=cfg.num_frame_stack, **common)
        self.bev_planner.load_state_dict(ck.get("bev") or synth.seeded_state_dict(self.bev_planner, prefix="uni.bev_planner."))
        self.bev_planner.to(self.device)
        self.branch_weights = torch.tensor(cfg.branch_weights).float().to(self.device)
        H = int((cfg.max_x - cfg.min_x) * cfg.pixels_per_meter)
        W = int((cfg.max_y - cfg.min_y) * cfg.pixels_per_meter)
        self.bev_center = [W / 2 + (cfg.min_y + cfg.max_y) / 2 * cfg.pixels_per_meter, H / 2 + (cfg.min_x + cfg.max_x) / 2 * cfg.pixels_per_meter]
        # ResNet.fc rides in the checkpoints but no forward uses it: without a gradient it would keep its DDP bucket
        # from ever becoming ready
        for p in self.bev_planner.bev_conv_emb[0].fc.parameters():
            p.requires_grad_(False)
        if what == "bev":
            self.bev_planner.train()
            self.bev_optim = optim.Adam([p for p in self.bev_planner.parameters() if p.requires_grad], lr=cfg.lr)
            self.bev_scheduler = StepLR(self.bev_optim, step_size=32, gamma=0.5)
            self.bev_ddp = _ddp(self.bev_planner, self.device)
            return
        num_input = (len(cfg.seg_channels) if cfg.point_painting else 0) + cfg.num_frame_stack + 10
        self.lidar_model = LiDARModel(num_input=num_input, num_features=cfg.num_features, backbone=cfg.backbone, min_x=cfg.min_x,
                                      max_x=cfg.max_x, min_y=cfg.min_y, max_y=cfg.max_y, pixels_per_meter=cfg.pixels_per_meter)
        self.lidar_model.load_state_dict(ck.get("lidar") or synth.seeded_state_dict(self.lidar_model, prefix="lidar."))
        self.bev_planner.eval()
        self.uniplanner = UniPlanner(self.bev_planner, num_input_feature=cfg.num_features[-1] * 6, **common)
        if ck.get("uniplanner"):
            self.uniplanner.load_state_dict(ck["uniplanner"])
        else:
            sd = synth.seeded_state_dict(self.uniplanner, prefix="uni.")
            sd.update({"bev_planner." + k: v for k, v in self.bev_planner.state_dict().items()})
            self.uniplanner.load_state_dict(sd)
        up = self.uniplanner
        # what lav_final_v2.py:72-84 hands to Adam; the teacher and the never-used *_other decoders take no gradient
        trainable = [up.plan_gru, up.plan_mlp, up.cast_grus_ego, up.cast_mlps_ego, up.cast_cmd_pred, up.lidar_conv_emb]
        for p in up.parameters():
            p.requires_grad_(False)
        params = []
        for m in trainable:
            for p in m.parameters():
                p.requires_grad_(True)
                params.append(p)
        for p in up.lidar_conv_emb[0].fc.parameters():
            p.requires_grad_(False)
        params = [p for p in params if p.requires_grad]
        if cfg.motion_only:
            for p in self.lidar_model.parameters():
                p.requires_grad_(False)
        else:
            params += list(self.lidar_model.parameters())
        self.student = _Student(self.lidar_model, self.uniplanner).to(self.device).train()
        self.bev_planner.eval()
        self.lidar_optim = optim.Adam(params, lr=cfg.lr)
        self.lidar_scheduler = StepLR(self.lidar_optim, step_size=4, gamma=0.5)
        self.det_criterion = DetLoss()
        self.seg_mask = build_seg_mask(h=H, w=W, cx=self.bev_center[0], cy=self.bev_center[1]).to(self.device)
        self.student_ddp = _ddp(self.student, self.device, find_unused=True)

    def state_dict(self, model_name):
        return {"bev": self.bev_planner, "lidar": getattr(self, "lidar_model", None),
                "uniplanner": getattr(self, "uniplanner", None)}[model_name].state_dict()

    # ------------------------------------------------------------------------------------------------------------
    def train_bev(self, bev, ego_locs, cmds, nxps, bras, locs, oris, typs, num_objs, other_weight=0.):
        cfg, d = self.cfg, self.device
        if not cfg.use_others_to_train:
            other_weight = 0.
        bev, ego_locs, nxps = bev.float().to(d), ego_locs.float().to(d), nxps.float().to(d)
        cmds, idxs = cmds.long().to(d), (1 - bras).bool().to(d)
        from .hipnn import use_precision
        # (round 6, first session: bf16x6 here - the fp16 pieces' measuring launches cost this launch-bound step what they saved; second
        # session: the fused BatchNorm launches leave the bounds, the measuring launches are gone: 44.3-44.6 ms per step against 48.4-56.8)
        with use_precision(cfg.conv_precision or "f16x3"):
            out = self.bev_ddp(bev, ego_locs, locs.float().to(d), oris.float().to(d), nxps, typs.to(d))
            loss, terms = bev_losses(out, ego_locs, cmds, idxs, cfg, self.branch_weights, other_weight)
            self.bev_optim.zero_grad()
            loss.backward()
        self.bev_optim.step()
        return _scalars(loss, terms)

    def train_lidar(self, lidars, num_points, heatmaps, sizemaps, orimaps, bev, ego_locs, cmds, nxps, bras, locs, oris, typs,
                    num_objs):
        cfg, d = self.cfg, self.device
        lidars, heatmaps, sizemaps, orimaps = lidars.to(d), heatmaps.to(d), sizemaps.to(d), orimaps.to(d)
        ego_locs, nxps, locs, oris, typs = ego_locs.float().to(d), nxps.float().to(d), locs.float().to(d), oris.float().to(d), typs.to(d)
        bev = bev.float().to(d)
        cmds, idxs = cmds.long().to(d), (1 - bras).bool().to(d)
        self.bev_planner.eval()
        from .hipnn import use_precision
        # round 6: the step's convolutions (forward, data gradient, weight gradient) on two fp16 pieces per operand (hipnn.train_precision)
        with use_precision(cfg.conv_precision or "f16x3"):
            lidar_out, uni_out = self.student_ddp(lidars, num_points, bev, ego_locs, locs, oris, nxps, typs)
            loss, terms = lidar_losses(self.det_criterion, lidar_out, uni_out, heatmaps, sizemaps, orimaps, bev[:, [0, 1, 2]],
                                       self.seg_mask, ego_locs, cmds, idxs, cfg, self.branch_weights)
            self.lidar_optim.zero_grad()
            loss.backward()
        self.lidar_optim.step()
        # The inference below only feeds the visualisation log.  It needs the inference engines re-packed from the
        # just-updated weights (~500 small copies), so it runs on the steps that are logged, not on all of them.
        # (round 6: enqueued BEFORE the loss terms are read back - its host work (eval(), the engines' refresh, ~100 launches) then runs
        # while the GPU is still in the backward pass instead of behind an empty queue)
        log = None
        if cfg.log_inference and self.steps % max(cfg.log_every, 1) == 0:
            log = self.mot_inference(lidars[0], num_points[0], cmds[0], nxps[0])
        info = _scalars(loss, terms)
        if log is not None:
            info.update(log)
        self.steps += 1
        return info

    @torch.no_grad()
    def mot_inference(self, lidar, num_point, cmd, nxp):
        """One eval-mode inference of sample 0 on the HIP inference kernels, as the reference does every step for its
        logs (lav_final_v2.py:228-236, 289-321)."""
        from ..model_inference import InferModel  # noqa: F401  (decode rules live there)
        self.student.eval()
        try:
            n = int(num_point)
            features, heat, size, ori, _ = self.lidar_model([lidar[:n]], [n])
            from .. import ops
            rows = ops.extract_peaks(heat[0], size[0], ori[0], apply_sigmoid=True).cpu().tolist()
            det = [(int(x), int(y), w, h, c, s) for sc, x, y, w, h, c, s in rows[1] if sc > 0.2]
            plan, _, other_locs, other_cmds = self.uniplanner.infer(features[0], det, int(cmd), nxp)
            return dict(ego_plan_locs=plan.cpu().numpy(), num_det=len(det))
        finally:
            self.student.train()
            self.bev_planner.eval()
