"""The GPU side of LAVAgent.run_step (team_code_v2/lav_agent_fast.py:233-323) as one object:

    half-sweep concat + ego-box removal -> ERFNet + softmax -> point painting -> 15-frame history ->
    temporal stacking in the current ego frame -> InferModel.forward -> brake net

Used by lav_amd.lav_agent.LAVAgent and by bench.py (the "full agent forward" of BASELINE.json).  Pose
bookkeeping (EKF loc/ori per frame) is handed in by the caller; nothing here touches the host except the
detection decode inside InferModel.det_inference.
"""
from __future__ import annotations

import math
from collections import deque

import numpy as np
import torch

from .model_inference import InferModel

GAP = 5  # NUM_REPEAT + 1 (lav_agent_fast.py:32-33)


def ego_box_mask(lidar: torch.Tensor) -> torch.Tensor:
    """True for points inside the ego-vehicle box (lav_agent_fast.py:450-452)."""
    x, y, z = lidar[:, 0], lidar[:, 1], lidar[:, 2]
    return (x > -2.4) & (x < 0) & (y > -0.8) & (y < 0.8) & (z > -1.5) & (z < -1)


def move_lidar_points(xyz: torch.Tensor, dloc, ori0: float, ori1: float) -> torch.Tensor:
    """Re-register a past sweep into the current ego frame (lav_agent_fast.py:547-565)."""
    dloc = np.asarray(dloc, np.float64) @ np.array([[math.cos(ori0), -math.sin(ori0)], [math.sin(ori0), math.cos(ori0)]])
    ori = ori1 - ori0
    R = torch.tensor([[math.cos(ori), math.sin(ori), 0.], [-math.sin(ori), math.cos(ori), 0.], [0., 0., 1.]],
                     dtype=torch.float32, device=xyz.device)
    out = xyz @ R
    out[:, 0] += float(dloc[0])
    out[:, 1] += float(dloc[1])
    return out


class FramePipeline:
    def __init__(self, lidar_model, uniplanner, seg_model, bra_model, camera_x=1.5, camera_z=2.4, num_frame_stack=2,
                 device=torch.device("cuda"), compact_ego_box: bool = False):
        self.device = device
        self.infer_model = InferModel(lidar_model, uniplanner, camera_x, camera_z, device=device)
        self.seg_model, self.bra_model = seg_model, bra_model
        self.num_frame_stack = num_frame_stack
        self.num_frame_keep = (num_frame_stack + 1) * GAP
        self.compact_ego_box = compact_ego_box
        self.reset()

    def reset(self):
        self.lidars, self.locs, self.oris = deque(), deque(), deque()
        self.prev_lidar = None

    def preprocess(self, lidar):
        """Ego-box removal.  Default: mark dropped points with x = NaN (they fail the pillar range test exactly
        like removed points, every downstream result is independent of point order and count) - no stream
        compaction, hence no device->host sync.  compact_ego_box=True reproduces the reference's boolean
        indexing literally."""
        m = ego_box_mask(lidar)
        if self.compact_ego_box:
            return lidar[~m]
        out = lidar.clone()
        out[:, 0] = torch.where(m, torch.full_like(out[:, 0], float("nan")), out[:, 0])
        return out

    def get_stacked_lidar(self):
        """lav_agent_fast.py:363-383: frames t, t-5, t-10 moved into the current frame + one-hot time."""
        loc0, ori0 = self.locs[-1], self.oris[-1]
        parts = []
        for i, t in enumerate(range(len(self.lidars) - 1, -1, -GAP)):
            lidar = self.lidars[t]
            xyz = move_lidar_points(lidar[:, :3], self.locs[t] - loc0, ori0, self.oris[t])
            onehot = torch.zeros((len(xyz), self.num_frame_stack + 1), dtype=xyz.dtype, device=xyz.device)
            onehot[:, i] = 1
            parts.append(torch.cat([xyz, lidar[:, 3:], onehot], dim=-1))
        return torch.cat(parts)

    @torch.no_grad()
    def step(self, lidar, all_rgbs, rgbs, tel_rgbs, loc, ori, nxps, cmd_value):
        """One frame.  lidar (n,4) f32, all_rgbs (3,3,288,256) f32, rgbs (1,3,288,768) f32, tel_rgbs (1,3,192,480)
        f32 - all resident in HBM; loc (2,) / ori host floats (EKF pose); nxps (2,) HBM; cmd_value int."""
        if self.prev_lidar is None:                      # first frame: only stash (lav_agent_fast.py:235-237)
            self.prev_lidar = lidar
            return None
        cur_lidar = self.preprocess(torch.cat([lidar, self.prev_lidar]))
        self.prev_lidar = lidar
        pred_sem = torch.softmax(self.seg_model(all_rgbs), dim=1)
        fused = self.infer_model.forward_paint(cur_lidar, pred_sem)
        self.lidars.append(fused)
        self.locs.append(np.asarray(loc, np.float64))
        self.oris.append(float(ori))
        if len(self.lidars) > self.num_frame_keep:
            self.lidars.popleft(); self.locs.popleft(); self.oris.popleft()
        lidar_points = self.get_stacked_lidar()
        out = self.infer_model(lidar_points, nxps, cmd_value)
        pred_bra = self.bra_model(rgbs, tel_rgbs)
        return dict(ego_embd=out[0], ego_plan_locs=out[1], ego_cast_locs=out[2], other_cast_locs=out[3],
                    other_cast_cmds=out[4], pred_bev=out[5], det=out[6], pred_bra=pred_bra, lidar_points=lidar_points)
