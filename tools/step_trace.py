#!/usr/bin/env python3
"""Where one GraphedFramePipeline.step spends its wall time: host marks without syncs (enqueue cost) and with a
device sync at every mark (phase durations).  GPU only."""
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402

dev = torch.device("cuda:0")
pipe, sds, _ = bench.build_pipeline(dev)
host, d = bench.synthetic_inputs(dev)
marks = {}
SYNC = False


def mark(name, t0):
    if SYNC:
        torch.cuda.synchronize()
    t = time.perf_counter()
    marks.setdefault(name, []).append(t - t0)
    return t


def step(self, lidar, all_rgbs, rgbs, tel_rgbs, loc, ori, nxps, cmd_value):
    t = time.perf_counter()
    n = min(int(lidar.shape[0]), self.P)
    self.b_tick[:n].copy_(lidar[:n], non_blocking=True)
    if n < self.P:
        self.b_tick[n:].fill_(float("nan"))
    self.b_all_rgbs.copy_(all_rgbs, non_blocking=True); self.b_rgbs.copy_(rgbs, non_blocking=True)
    self.b_tel.copy_(tel_rgbs, non_blocking=True); self.b_nxp.copy_(nxps, non_blocking=True)
    t = mark("1 input copies", t)
    self.poses.append((np.asarray(loc, np.float64), float(ori)))
    if len(self.poses) > self.num_frame_keep:
        self.poses.popleft()
    self._set_pose_buffers()
    t = mark("2 pose buffers", t)
    main = torch.cuda.current_stream()
    self.ev_in.record(main)
    o_lidar = self._replay("lidar", self._g_lidar, self.s_cap)
    t = mark("3 lidar graph", t)
    self.s_bra.wait_event(self.ev_in)
    with torch.cuda.stream(self.s_bra):
        o_bra = self._replay("brake", self._g_brake, self.s_bra)
    t = mark("4 brake graph", t)
    self.ev_feat.record(main)
    self.s_ego.wait_event(self.ev_feat)
    with torch.cuda.stream(self.s_ego):
        o_ego = self._replay(("ego", cmd_value), self._g_ego, self.s_ego, cmd_value)
    t = mark("5 ego graph", t)
    o_heads = self._replay("heads", self._g_heads, self.s_cap)
    t = mark("6 heads graph", t)
    self.frame_no += 1
    det_rows = o_heads["det_raw"].cpu().tolist()
    t = mark("7 det_raw D2H (sync)", t)
    det = self._decode(det_rows)
    up = self.infer_model.uniplanner
    H, W = self.b_features.size(2) * 2, self.b_features.size(3) * 2
    locs, oris = up.others_from_detections(det[1], H, W)
    N = min(len(locs), 15)
    t = mark("8 host decode", t)
    self.b_locs[:N].copy_(torch.tensor(locs[:N], dtype=torch.float32), non_blocking=True)
    self.b_oris[:N].copy_(torch.tensor(oris[:N], dtype=torch.float32), non_blocking=True)
    ob = self._replay(("others", N), self._g_others, self.s_cap, N)
    t = mark("9 others graph", t)
    main.wait_stream(self.s_ego)
    main.wait_stream(self.s_bra)
    t = mark("10 joins", t)


for i in range(25):
    loc, ori = bench.pose(i)
    pipe.step(d["ticks"][i % 4], d["all_rgbs"], d["rgbs"], d["tel_rgbs"], loc, ori, d["nxp"], 3)
torch.cuda.synchronize()
for SYNC in (False, True):
    marks.clear()
    t0 = time.perf_counter()
    for i in range(25, 75):
        loc, ori = bench.pose(i)
        step(pipe, d["ticks"][i % 4], d["all_rgbs"], d["rgbs"], d["tel_rgbs"], loc, ori, d["nxp"], 3)
    torch.cuda.synchronize()
    print(f"--- sync at every mark: {SYNC}; frame {(time.perf_counter() - t0) / 50 * 1e3:.3f} ms")
    for k, v in marks.items():
        print(f"  {k:24s} {np.mean(v[5:]) * 1e6:8.1f} us")
