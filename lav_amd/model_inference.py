"""InferModel / CoordConverter with the reference's surface (team_code_v2/model_inference.py:14-297):

    InferModel(lidar_model, uniplanner, camera_x, camera_z, device)
    .forward_paint(cur_lidar (N,4), pred_sem (3,5,288,256)) -> (N,8)
    .forward(lidar_points (M,11), nxps (2,), cmd_value) ->
        (ego_embd, ego_plan_locs, ego_cast_locs, other_cast_locs, other_cast_cmds, pred_bev, det)

Point painting, pillar scatter, the BEV convolutions, peak extraction, the rotated crops, the ResNet embedder and the
GRU decoders all run on liblav_amd; what is left to torch are trivial elementwise / tiny ops (sigmoid of cast_cmd_pred,
transform_points, max / average pooling of the embedder).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn
from torch.nn import functional as F

from . import ops
from .planner_common import crop_feature, transform_points  # noqa: F401  (re-exported like the reference module)

CAMERA_YAWS = [-60, 0, 60]


def carla_matrix(x, y, z, yaw_deg=0.0):
    """carla.Transform(Location(x,y,z), Rotation(yaw=yaw)).get_matrix() for roll = pitch = 0, in float32 as
    LibCarla computes it (geom/Transform.h).  CARLA itself is not needed to build the projection."""
    f = np.float32
    a = f(yaw_deg) * (f(np.pi) / f(180.0))
    c, s = np.cos(a, dtype=f), np.sin(a, dtype=f)
    return np.array([[c, -s, 0, x], [s, c, 0, y], [0, 0, 1, z], [0, 0, 0, 1]], f)


def carla_inverse_matrix(x, y, z, yaw_deg=0.0):
    f = np.float32
    a = f(yaw_deg) * (f(np.pi) / f(180.0))
    c, s = np.cos(a, dtype=f), np.sin(a, dtype=f)
    ax, ay, az = -f(x), -f(y), -f(z)
    tx = ax * c + ay * s + az * f(0)
    ty = ax * (-s) + ay * c + az * f(-0.0)
    tz = ax * f(-0.0) + ay * f(0) + az * f(1)
    return np.array([[c, s, 0, tx], [-s, c, 0, ty], [0, 0, 1, tz], [0, 0, 0, 1]], f)


class CoordConverter(nn.Module):
    """LiDAR -> image-plane integer coordinates for one camera (model_inference.py:255-297).  Holds K,
    lidar_to_world, world_to_cam as parameters like the reference; forward() runs the paint kernel's
    projection and returns (N,3) int64 (u, v, depth) truncated toward zero."""

    def __init__(self, cam_yaw, lidar_xyz=(0, 0, 2.5), cam_xyz=(1.4, 0, 2.5), rgb_h=320, rgb_w=320, fov=60):
        super().__init__()
        focal = rgb_w / (2.0 * np.tan(fov * np.pi / 360.0))
        K = torch.eye(3)
        K[0, 0] = K[1, 1] = focal
        K[0, 2] = rgb_w / 2.0
        K[1, 2] = rgb_h / 2.0
        self.K = nn.Parameter(K, requires_grad=False)
        self.lidar_to_world = nn.Parameter(torch.from_numpy(carla_matrix(*lidar_xyz)), requires_grad=False)
        self.world_to_cam = nn.Parameter(torch.from_numpy(carla_inverse_matrix(*cam_xyz, yaw_deg=cam_yaw)), requires_grad=False)
        self.rgb_h, self.rgb_w = rgb_h, rgb_w

    def matrices(self):
        return (self.K.detach().cpu().numpy(), self.lidar_to_world.detach().cpu().numpy(),
                self.world_to_cam.detach().cpu().numpy())

    def forward(self, lidar):
        sem = torch.zeros((1, 5, self.rgb_h, self.rgb_w), dtype=torch.float32, device=lidar.device)
        _, uvz = ops.paint(lidar, sem, ops.make_cameras([self.matrices()]), want_uvz=True)
        return uvz[0].long()


def extract_peak(heatmap, max_pool_ks: int = 7, min_score: float = 0.1, max_det: int = 15, break_tie: bool = False):
    """The reference's module-level helper (model_inference.py:189-202): 7x7 max-pool NMS + top-`max_det` of one (H,W) heat
    map -> [(score, x, y), ...] with score > min_score, on lav_extract_peaks (one launch, one small device->host copy;
    ties between equal scores are ordered by pixel index, deterministically, instead of topk's unspecified order)."""
    if break_tie:
        heatmap = heatmap + 1e-7 * torch.randn(*heatmap.size(), device=heatmap.device)
    h, w = heatmap.shape
    z = heatmap.new_zeros((1, h, w))
    rows = ops.extract_peaks(heatmap[None].contiguous(), z, z, ks=max_pool_ks, max_det=min(max_det, h * w)).cpu()
    return [(float(s), int(x), int(y)) for s, x, y in rows[0, :, :3].tolist() if s > min_score]


class InferModel(nn.Module):
    def __init__(self, lidar_model, uniplanner, camera_x, camera_z, device=torch.device("cuda"), precision=None):
        """precision: lav_conv.precision of the eval engines this wrapper runs its modules at (default ops.frame_precision():
        LAV_CONV_F16X3 - every split-kernel convolution on two fp16 pieces, the scale handed from layer to layer; waypoints stay
        within 1e-4 of the reference, tests/test_gpu_e2e.py).  The engines are cached per precision on the modules."""
        super().__init__()
        self.precision = ops.frame_precision() if precision is None else int(precision)
        self.lidar_model = lidar_model
        self.uniplanner = uniplanner
        self.coord_converters = [CoordConverter(yaw, lidar_xyz=[0, 0, camera_z], cam_xyz=[camera_x, 0, camera_z],
                                                rgb_h=288, rgb_w=256, fov=64) for yaw in CAMERA_YAWS]
        self._cams = ops.make_cameras([cc.matrices() for cc in self.coord_converters])
        self.pixels_per_meter = uniplanner.pixels_per_meter
        self.offset_x, self.offset_y = uniplanner.offset_x, uniplanner.offset_y
        self.crop_size = uniplanner.crop_size
        self.num_cmds, self.num_plan = uniplanner.num_cmds, uniplanner.num_plan
        self.plan, self.cast, self.cast_cmd_pred = uniplanner.plan, uniplanner.cast, uniplanner.cast_cmd_pred
        ny, nx = lidar_model.point_pillar_net.ny, lidar_model.point_pillar_net.nx
        self._bev_hw = (int(ny), int(nx))     # size of the head maps the detections live on

    @torch.no_grad()
    def forward_paint(self, cur_lidar, pred_sem):
        """(N,4) + softmax maps (3,5,288,256) -> (N,8) = cat(lidar, painted); one kernel for the class-0
        suppression, the 3 projections, the gather and the concat (model_inference.py:44-50,75-93)."""
        return ops.paint(cur_lidar, pred_sem, self._cams)

    @torch.no_grad()
    def forward(self, lidar_points, nxps, cmd_value):
        lm = self.lidar_model
        with ops.precision(self.precision):
            canvas = lm.point_pillar_net([lidar_points], [len(lidar_points)])
            features = lm.backbone(canvas)
            heat, size, ori, pred_bev = lm.heads(features)
            det = self.det_decode(ops.extract_peaks(heat[0], size[0], ori[0], apply_sigmoid=True).cpu().tolist())
            ego_embd, ego_plan, ego_cast, other_cast, other_cmds = self.uniplanner.infer_all(features[0], det[1], cmd_value, nxps, amax=ops.amax_of(features))
        return ego_embd, ego_plan, ego_cast, other_cast, other_cmds, pred_bev, det

    def det_inference(self, heatmaps, sizemaps, orimaps, min_score=0.2):
        """Peaks -> [(x, y, w, h, cos, sin)] per class with the reference's score/size/range filters
        (model_inference.py:95-121): one lav_extract_peaks launch and one device->host copy for all classes."""
        return self.det_decode(ops.extract_peaks(heatmaps, sizemaps, orimaps).cpu().tolist(), min_score)

    def det_decode(self, det_rows, min_score=0.2):
        """Host half of det_inference: the score / size / range filters on (ncls, 15, 7) rows."""
        dets = []
        for i, rows in enumerate(det_rows):
            det = []
            for s, x, y, w, h, cos, sin in rows:
                if not s > min_score:
                    continue
                x, y = int(x), int(y)
                if i == 1 and max(w, h) < 0.1 * self.pixels_per_meter:
                    continue
                dist = np.linalg.norm([x - 160, y - 280])  # reference hard-codes the ego pixel (:113)
                if dist <= 2 or dist >= 30 * self.pixels_per_meter:
                    continue
                det.append((x, y, w, h, cos, sin))
            dets.append(det)
        return dets

    def det_decode_fast(self, rows: np.ndarray, min_score=0.2):
        """det_decode + UniPlanner.others_from_detections on a (ncls, 15, 7) float32 array with numpy masks instead of
        Python loops over the rows (the frame's critical chain waits on this): returns (dets, locs (N,2), oris (N,))
        identical to the loop versions."""
        ppm = self.pixels_per_meter
        rows64 = rows.astype(np.float64)     # the loops compare Python floats: float32 values widened, thresholds in float64
        s, x, y = rows64[..., 0], rows[..., 1].astype(np.int64), rows[..., 2].astype(np.int64)
        w, h = rows64[..., 3], rows64[..., 4]
        dist = np.sqrt(((x - 160) ** 2 + (y - 280) ** 2).astype(np.float64))
        keep = (s > min_score) & (dist > 2) & (dist < 30 * ppm)
        if rows.shape[0] > 1:
            keep[1] &= ~(np.maximum(w[1], h[1]) < 0.1 * ppm)
        dets = [[(int(x[i, j]), int(y[i, j]), float(w[i, j]), float(h[i, j]), float(rows[i, j, 5]), float(rows[i, j, 6]))
                 for j in np.flatnonzero(keep[i])] for i in range(rows.shape[0])]
        up = self.uniplanner
        ox, oy = up.offsets()
        H = W = None
        if rows.shape[0] > 1 and keep[1].any():
            H, W = self._bev_hw
            cx, cy = float(W / 2 + ox * W / 2), float(H / 2 + oy * H / 2)
            j = np.flatnonzero(keep[1])
            X, Y = x[1, j].astype(np.float64), y[1, j].astype(np.float64)
            far = np.sqrt((X - cx) ** 2 + (Y - cy) ** 2) > 4
            j, X, Y = j[far], X[far], Y[far]
            locs = np.stack([(X - cx) / up.pixels_per_meter, (Y - cy) / up.pixels_per_meter], axis=1)
            oris = np.arctan2(rows[1, j, 6].astype(np.float64), rows[1, j, 5].astype(np.float64))
            return dets, locs, oris
        return dets, np.zeros((0, 2)), np.zeros((0,))

    def uniplanner_infer(self, features, det, cmd_value, nxp):
        with ops.precision(self.precision):
            return self.uniplanner.infer_all(features, det, cmd_value, nxp)

    def point_painting(self, lidar, sems):
        """Painted channels only, for already class-0-suppressed maps `sems` (3,C,H,W) - the reference's
        helper signature (model_inference.py:75-93).  Implemented by prepending a zero class-0 plane."""
        sem5 = torch.cat([torch.zeros_like(sems[:, :1]), sems], dim=1)
        return ops.paint(lidar, sem5, self._cams)[:, lidar.shape[1]:]
