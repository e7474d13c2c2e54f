"""UniPlanner with the reference's constructor, state_dict keys and inference methods
(team_code_v2/models/uniplanner.py:8-53, 180-352): ResNet-18 embedding of rotated 96x96 feature crops on the
MFMA convolution, multi-modal cast GRUs and the iterative plan GRU on liblav_amd's GRU kernels.
"""
from __future__ import annotations

import numpy as np
import torch
from torch import nn

from . import ops
from .lidar import _Engine
from .planner_common import _hip_train, DecoderMixin, crop_feature, crop_feature_torch, sample_others, transform_points
from .resnet import resnet18


class UniPlanner(DecoderMixin, _Engine):
    def __init__(self, bev_planner, pixels_per_meter=2, crop_size=64, x_offset=0, y_offset=0.75, feature_x_jitter=1,
                 feature_angle_jitter=10, num_plan=10, k=16, num_input_feature=96, num_out_feature=64, num_cmds=6,
                 max_num_cars=4, num_plan_iter=1):
        super().__init__()
        self.num_cmds, self.num_plan, self.num_plan_iter = num_cmds, num_plan, num_plan_iter
        self.max_num_cars = max_num_cars
        self.bev_planner = bev_planner
        self.num_out_feature = num_out_feature
        self.pixels_per_meter, self.crop_size = pixels_per_meter, crop_size
        self.feature_x_jitter = feature_x_jitter
        self.feature_angle_jitter = np.deg2rad(feature_angle_jitter)
        self.offset_x = nn.Parameter(torch.tensor(x_offset).float(), requires_grad=False)
        self.offset_y = nn.Parameter(torch.tensor(y_offset).float(), requires_grad=False)
        self.lidar_conv_emb = nn.Sequential(resnet18(num_channels=num_input_feature), nn.AdaptiveAvgPool2d((1, 1)),
                                            nn.Flatten())
        self.plan_gru = nn.GRU(4, 512, batch_first=True)
        self.plan_mlp = nn.Linear(512, 2)
        self.cast_grus_ego = nn.ModuleList([nn.GRU(512, 64, batch_first=True) for _ in range(num_cmds)])
        self.cast_mlps_ego = nn.ModuleList([nn.Linear(64, 2) for _ in range(num_cmds)])
        # present in the reference's checkpoints but never used by any forward (uniplanner.py:296-300)
        self.cast_grus_other = nn.ModuleList([nn.GRU(512, 64, batch_first=True) for _ in range(num_cmds)])
        self.cast_mlps_other = nn.ModuleList([nn.Linear(64, 2) for _ in range(num_cmds)])
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

    def train(self, mode: bool = True):
        """The privileged teacher rides along as a submodule but is frozen: it stays in eval mode (the reference re-applies
        bev_planner.eval() at every forward, uniplanner.py:58) and is skipped when the mode of the rest is switched, which
        also keeps its packed inference engines alive across the trainer's train()/eval() toggles."""
        if mode != self.training:
            self._drop()
        self.training = mode
        for name, child in self.named_children():
            if name != "bev_planner":
                child.train(mode)
        self.bev_planner.eval()
        return self

    def _cast_modules(self):
        return self.cast_grus_ego, self.cast_mlps_ego

    def crop_feature(self, features, rel_locs, rel_oris, pixels_per_meter=4, crop_size=96, map_index=None, amax=None):
        """map_index (int32, per crop): take crop i from features[map_index[i]] instead of features[i] - the training
        forwards crop several vehicles out of each sample's map without materialising one copy of the map per vehicle.
        amax (eval): the ops.Amax of the feature map - a bilinear crop never exceeds the map's largest magnitude, so the crops carry
        it on to the stem convolution (LAV_CONV_F16X3's scale, lav_conv2d_amax)."""
        if amax is not None and not self.training and map_index is None:
            ox, oy = self.offsets()
            crops = crop_feature(features, rel_locs, rel_oris, pixels_per_meter, crop_size, ox, oy)
            crops._lav_amax = amax
            return crops
        ox, oy = self.offsets()
        if map_index is not None and features.is_cuda and (_hip_train("CROP") or not self.training):   # HIP forward + backward (autograd.Function)
            return ops.crop_rotate_indexed(features, map_index, rel_locs, rel_oris, pixels_per_meter, crop_size, ox, oy)
        if map_index is not None:
            features = features[map_index.long()]
        if self.training:   # autograd path (affine_grid + grid_sample)
            return crop_feature_torch(features, rel_locs, rel_oris, pixels_per_meter, crop_size, ox, oy)
        return crop_feature(features, rel_locs, rel_oris, pixels_per_meter, crop_size, ox, oy)

    def embed_cast(self, crops, oris=None, locs=None, want_cmds=False):
        """Eval-mode tail of both branches in two steps instead of ~17 launches: the ResNet-18 trunk on the crops, then ONE
        lav_embed_cast launch = AdaptiveAvgPool2d + Flatten (uniplanner.py:36-40), the six cast GRUs (:288-308), cast_cmd_pred
        (:50-53) and - for the other vehicles - transform_points + translate into the ego frame (model_inference.py:164-165).
        crops (B,C,crop,crop) -> (embd (B,512), cast (B,num_cmds,T,2), cmds (B,num_cmds) or None)."""
        fmap = self.lidar_conv_emb[0](crops)
        w = self._dec(crops.device)["cast"]
        lin = self.cast_cmd_pred[0]
        return ops.embed_cast(fmap, w["w_ih"], w["w_hh"], w["b_ih"], w["b_hh"], w["mlp_w"], w["mlp_b"], self.num_plan,
                              cmd_w=lin.weight if want_cmds else None, cmd_b=lin.bias if want_cmds else None, oris=oris, locs=locs)

    def others_from_detections(self, det, H, W):
        """Pixel detections -> ego-frame metres and headings, skipping the ego's own box
        (uniplanner.py:194-212 / model_inference.py:125-144)."""
        ox, oy = self.offsets()
        cx = float(W / 2 + ox * W / 2)
        cy = float(H / 2 + oy * H / 2)
        locs, oris = [], []
        for X, Y, h, w, cos, sin in det:
            if np.linalg.norm([X - cx, Y - cy]) <= 4:
                continue
            locs.append([(X - cx) / self.pixels_per_meter, (Y - cy) / self.pixels_per_meter])
            oris.append(float(np.arctan2(sin, cos)))
        return locs, oris

    def forward(self, features, bev, ego_locs, locs, oris, nxps, typs):
        """Training forward (uniplanner.py:56-150): the student decodes from jittered crops of the LiDAR feature map,
        the frozen privileged BEVPlanner (eval mode, no_grad - it runs on the HIP inference kernels) from the same crops
        of the ground-truth BEV.  features (B,384,160,160), bev (B,9,320,320), ego_locs (B,T+1,2), locs (B,N+1,T+1,2),
        oris (B,N+1), nxps (B,2), typs (B,N+1)."""
        self.bev_planner.eval()
        teacher = self.bev_planner
        ppm, crop = self.pixels_per_meter, self.crop_size
        pick, N = sample_others(self, ego_locs, locs, oris, typs)
        if pick is not None:
            other_embd = self.lidar_conv_emb(self.crop_feature(features, pick["crop_locs"], pick["crop_oris"], ppm / 2, crop,
                                                               map_index=pick["sample"]))
            other_locs = pick["other_locs"]
            other_cast_cmds = self.cast_cmd_pred(other_embd)
            with torch.no_grad():
                t_embd = teacher.bev_conv_emb(teacher.crop_feature(bev, pick["crop_locs"], pick["crop_oris"], ppm, crop * 2,
                                                                   map_index=pick["sample"]))
                other_cast_locs_expert, other_cast_cmds_expert = teacher.cast(t_embd), teacher.cast_cmd_pred(t_embd)
        else:
            z = dict(dtype=features.dtype, device=features.device)
            other_locs = torch.zeros((N, self.num_plan, 2), **z)
            other_cast_locs = torch.zeros((N, self.num_cmds, self.num_plan, 2), **z)
            other_cast_cmds = torch.zeros((N, self.num_cmds), **z)
            other_cast_locs_expert, other_cast_cmds_expert = torch.zeros_like(other_cast_locs), torch.zeros_like(other_cast_cmds)
            other_embd = None
        B = features.size(0)
        locs_jitter = (torch.rand((B, 2)) * 2 - 1).float().to(locs.device) * self.feature_x_jitter
        locs_jitter[:, 1] = 0
        oris_jitter = (torch.rand((B,)) * 2 - 1).float().to(oris.device) * self.feature_angle_jitter
        ego_locs = transform_points(ego_locs[:, 1:] - locs_jitter[:, None], -oris_jitter)
        nxps = transform_points(nxps[:, None] - locs_jitter[:, None], -oris_jitter)[:, 0]
        every = torch.arange(B, dtype=torch.int32, device=features.device)
        ego_embd = self.lidar_conv_emb(self.crop_feature(features, locs_jitter, oris_jitter, ppm / 2, crop, map_index=every))
        with torch.no_grad():
            t_embd = teacher.bev_conv_emb(teacher.crop_feature(bev, locs_jitter, oris_jitter, ppm, crop * 2, map_index=every))
            ego_cast_locs_expert = teacher.cast(t_embd)
            ego_plan_locs_expert = teacher.plan(t_embd, nxps, cast_locs=ego_cast_locs_expert, pixels_per_meter=ppm, crop_size=crop * 2)
        # the reference decodes others and ego with separate cast() calls on the same *_ego GRUs (uniplanner.py:296-300):
        # one call over the concatenated embeddings is the same arithmetic in half the GRU launches
        if other_embd is not None:
            both = self.cast(torch.cat([other_embd, ego_embd]), mode="ego")
            other_cast_locs, ego_cast_locs = both[:other_embd.size(0)], both[other_embd.size(0):]
        else:
            ego_cast_locs = self.cast(ego_embd, mode="ego")
        ego_plan_locs = self.plan(ego_embd, nxps, cast_locs=ego_cast_locs, pixels_per_meter=ppm, crop_size=crop * 2)
        return (other_locs, other_cast_locs, other_cast_cmds, other_cast_locs_expert, other_cast_cmds_expert,
                ego_locs, ego_plan_locs, ego_cast_locs, self.cast_cmd_pred(ego_embd), ego_cast_locs_expert, ego_plan_locs_expert)

    @torch.no_grad()
    def infer(self, features, det, cmd, nxp):
        """features (384,160,160), det list of (X,Y,h,w,cos,sin), cmd int, nxp (2,)
        -> ego_plan_locs (T,2), ego_cast_locs (T,2), other_cast_locs (N,6,T,2), other_cast_cmds (N,6)."""
        ego_embd, plan, cast, oc, om = self.infer_all(features, det, cmd, nxp)
        return plan, cast, oc, om

    @torch.no_grad()
    def infer_all(self, features, det, cmd, nxp, amax=None):
        """amax: the feature map's ops.Amax when the backbone left one (InferModel hands it on)."""
        dev = features.device
        H, W = features.size(1) * 2, features.size(2) * 2
        locs, oris = self.others_from_detections(det, H, W)
        N = len(locs)
        ppm_f = self.pixels_per_meter / 2
        if N > 0:
            locs_t = torch.tensor(locs, dtype=torch.float32, device=dev)
            oris_t = torch.tensor(oris, dtype=torch.float32, device=dev)
            crops = self.crop_feature(features.expand(N, *features.size()), locs_t, oris_t, ppm_f, self.crop_size, amax=amax)
            _, other_cast, other_cmds = self.embed_cast(crops, oris=oris_t, locs=locs_t, want_cmds=True)
        else:  # the reference returns CPU zeros here (model_inference.py:167-168) - keep
            other_cast = torch.zeros((0, self.num_cmds, self.num_plan, 2))
            other_cmds = torch.zeros((0, self.num_cmds))
        ego_crop = self.crop_feature(features[None], features.new_zeros((1, 2)), features.new_zeros((1,)), ppm_f, self.crop_size, amax=amax)
        ego_embd, ego_cast, _ = self.embed_cast(ego_crop)
        # only the commanded branch of the plan GRU is needed: branches never interact (uniplanner.py:264-275)
        ego_plan = self.plan(ego_embd, nxp[None], cast_locs=ego_cast, pixels_per_meter=self.pixels_per_meter,
                             crop_size=self.crop_size * 2, cmd=int(cmd))[0, -1, 0]
        return ego_embd, ego_plan, ego_cast[0, int(cmd)], other_cast, other_cmds
