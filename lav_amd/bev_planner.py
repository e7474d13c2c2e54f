"""BEVPlanner (privileged teacher) with the reference's constructor and state_dict keys
(lav/models/bev_planner_v2.py:7-71).  It rides inside UniPlanner checkpoints (`bev_planner.*`), so it must
exist for drop-in loading; its cast/plan decoders run on the same HIP GRU kernels.
"""
from __future__ import annotations

import numpy as np
import torch
from torch import nn

from . import ops
from .lidar import _Engine
from .planner_common import _hip_train, DecoderMixin, crop_feature, crop_feature_torch, sample_others
from .resnet import resnet18


class BEVPlanner(DecoderMixin, _Engine):
    def __init__(self, pixels_per_meter=2, crop_size=64, x_offset=0, y_offset=0.75, feature_x_jitter=1,
                 feature_angle_jitter=10, num_plan=10, k=16, num_out_feature=64, num_cmds=6, max_num_cars=5,
                 num_plan_iter=1, num_frame_stack=0):
        super().__init__()
        self.num_cmds, self.num_plan, self.num_plan_iter = num_cmds, num_plan, num_plan_iter
        self.max_num_cars, self.num_out_feature = max_num_cars, num_out_feature
        self.pixels_per_meter, self.crop_size = pixels_per_meter, crop_size
        self.feature_x_jitter = feature_x_jitter
        self.feature_angle_jitter = np.deg2rad(feature_angle_jitter)
        self.offset_x = nn.Parameter(torch.tensor(x_offset).float(), requires_grad=False)
        self.offset_y = nn.Parameter(torch.tensor(y_offset).float(), requires_grad=False)
        self.bev_conv_emb = nn.Sequential(resnet18(num_channels=3 + 2 * (num_frame_stack + 1)),
                                          nn.AdaptiveAvgPool2d((1, 1)), nn.Flatten())
        self.plan_gru = nn.GRU(4, 512, batch_first=True)
        self.plan_mlp = nn.Linear(512, 2)
        self.cast_grus = nn.ModuleList([nn.GRU(512, 64, batch_first=True) for _ in range(num_cmds)])
        self.cast_mlps = nn.ModuleList([nn.Linear(64, 2) for _ in range(num_cmds)])
        self.cast_cmd_pred = nn.Sequential(nn.Linear(512, num_cmds), nn.Sigmoid())
        self._drop()

    def _drop(self):
        super()._drop()
        self._drop_dec()
        object.__setattr__(self, "_offsets", None)

    def _mark_stale(self):
        """Parameters may have changed in place (train()/eval() toggle, load_state_dict): the conv engines re-pack themselves at
        their next use; the stacked decoder weights and the cached offsets are plain copies, so they are dropped and rebuilt."""
        super()._mark_stale()
        self._drop_dec()
        object.__setattr__(self, "_offsets", None)

    def offsets(self):
        """(offset_x, offset_y) as host floats, read from the parameters once (no device->host sync per frame;
        keeps the forward HIP-graph capturable)."""
        if getattr(self, "_offsets", None) is None:
            object.__setattr__(self, "_offsets", (float(self.offset_x), float(self.offset_y)))
        return self._offsets

    def _cast_modules(self):
        return self.cast_grus, self.cast_mlps

    def crop_feature(self, features, rel_locs, rel_oris, pixels_per_meter=4, crop_size=96, map_index=None):
        """map_index (int32, per crop): take crop i from features[map_index[i]] instead of features[i] - the training
        forwards crop several vehicles out of each sample's map without materialising one copy of the map per vehicle."""
        ox, oy = self.offsets()
        if map_index is not None and features.is_cuda and (_hip_train("CROP") or not self.training):   # HIP forward + backward (autograd.Function)
            return ops.crop_rotate_indexed(features, map_index, rel_locs, rel_oris, pixels_per_meter, crop_size, ox, oy)
        if map_index is not None:
            features = features[map_index.long()]
        if self.training:   # autograd path (affine_grid + grid_sample)
            return crop_feature_torch(features, rel_locs, rel_oris, pixels_per_meter, crop_size, ox, oy)
        return crop_feature(features, rel_locs, rel_oris, pixels_per_meter, crop_size, ox, oy)

    @torch.no_grad()
    def infer(self, bev, nxps):
        """bev_planner_v2.py:46-70."""
        crop = self.crop_feature(bev, bev.new_zeros((1, 2)), bev.new_zeros((1,)), self.pixels_per_meter, self.crop_size * 2)
        embd = self.bev_conv_emb(crop)
        cast = self.cast(embd)
        plan = self.plan(embd, nxps, cast_locs=cast, pixels_per_meter=self.pixels_per_meter, crop_size=self.crop_size * 2)
        return plan, cast, self.cast_cmd_pred(embd)

    def forward(self, bev, ego_locs, locs, oris, nxps, typs):
        """Training forward of the privileged planner (bev_planner_v2.py:72-174): forecasts for up to `max_num_cars`
        jittered crops around other vehicles and the ego plan from the un-jittered ego crop.
        bev (B,9,320,320), ego_locs (B,T+1,2), locs (B,N+1,T+1,2), oris (B,N+1), nxps (B,2), typs (B,N+1)."""
        pick, N = sample_others(self, ego_locs, locs, oris, typs)
        ppm, crop = self.pixels_per_meter, self.crop_size * 2
        if pick is not None:
            other_embd = self.bev_conv_emb(self.crop_feature(bev, pick["crop_locs"], pick["crop_oris"], ppm, crop, map_index=pick["sample"]))
            other_locs = pick["other_locs"]
            other_cast_cmds = self.cast_cmd_pred(other_embd)
        else:
            z = dict(dtype=bev.dtype, device=bev.device)
            other_locs = torch.zeros((N, self.num_plan, 2), **z)
            other_cast_locs = torch.zeros((N, self.num_cmds, self.num_plan, 2), **z)
            other_cast_cmds = torch.zeros((N, self.num_cmds), **z)
            other_embd = None
        B = bev.size(0)
        ego_embd = self.bev_conv_emb(self.crop_feature(bev, bev.new_zeros((B, 2)), bev.new_zeros((B,)), ppm, crop,
                                                       map_index=torch.arange(B, dtype=torch.int32, device=bev.device)))
        if other_embd is not None:     # one cast() over others + ego: same arithmetic, half the GRU launches
            both = self.cast(torch.cat([other_embd, ego_embd]))
            other_cast_locs, ego_cast_locs = both[:other_embd.size(0)], both[other_embd.size(0):]
        else:
            ego_cast_locs = self.cast(ego_embd)
        ego_plan_locs = self.plan(ego_embd, nxps, cast_locs=ego_cast_locs, pixels_per_meter=ppm, crop_size=crop)
        return other_locs, other_cast_locs, other_cast_cmds, ego_plan_locs, ego_cast_locs, self.cast_cmd_pred(ego_embd)
