"""Oracle: one full agent frame on the CPU (TEST INFRASTRUCTURE - see oracle/__init__.py).

Restates the GPU part of LAVAgent.run_step (team_code_v2/lav_agent_fast.py:233-323) by chaining the oracle
pieces: half-sweep concat + ego-box removal (:240-247, 450-457), ERFNet + softmax (:263-264), point painting
(:266), temporal stacking (:277, 363-383), InferModel.forward (:317) and the brake net (:323).  The two camera
networks are evaluated by oracle/camera.py's torch restatement over the modules' parameters.  This is what
bench.py times as `cpu_baseline` (kind "port").
"""
from __future__ import annotations

import numpy as np
import torch

from . import bev as obev
from . import camera as ocam
from . import paint as opaint
from . import pillar as opillar


def stack(lidars, locs, oris, gap=5, num_frame_stack=2):
    loc0, ori0 = locs[-1], oris[-1]
    parts = []
    for i, t in enumerate(range(len(lidars) - 1, -1, -gap)):
        l = torch.from_numpy(lidars[t])
        xyz = obev.move_lidar_points(l[:, :3], locs[t] - loc0, ori0, oris[t])
        onehot = torch.zeros((len(xyz), num_frame_stack + 1))
        onehot[:, i] = 1
        parts.append(torch.cat([xyz, l[:, 3:], onehot], dim=-1))
    return torch.cat(parts).numpy()


@torch.no_grad()
def frame(lidar_tick, prev_tick, history, all_rgbs, rgbs, tel_rgbs, seg_cpu, bra_cpu, lsd, usd, pn_sd, nxp, cmd,
          loc, ori):
    """history: dict(lidars=[...], locs=[...], oris=[...]) mutated like the agent's deques."""
    cur = opaint.preprocess(np.concatenate([lidar_tick, prev_tick]))
    sem = torch.softmax(ocam.seg_forward(seg_cpu, all_rgbs), dim=1).numpy()
    fused = opaint.forward_paint(cur, sem)
    history["lidars"].append(fused); history["locs"].append(np.asarray(loc, np.float64)); history["oris"].append(float(ori))
    for k in ("lidars", "locs", "oris"):
        del history[k][:-15]
    pts = stack(history["lidars"], history["locs"], history["oris"])
    canvas = torch.from_numpy(opillar.pillar_forward([pts], [len(pts)], pn_sd)["canvas"])
    feat = obev.conv_backbone(canvas, lsd)
    heat, size, orim, seg = obev.lidar_heads(feat, lsd)
    det = obev.det_inference(torch.sigmoid(heat[0]), size[0], orim[0])
    e, p, c, oc, om = obev.uniplanner_infer(feat[0], det[1], cmd, nxp, usd)
    bra = ocam.brake_forward(bra_cpu, rgbs, tel_rgbs)
    return dict(ego_plan_locs=p, ego_cast_locs=c, other_cast_locs=oc, other_cast_cmds=om, pred_bev=seg, det=det,
                pred_bra=bra)
