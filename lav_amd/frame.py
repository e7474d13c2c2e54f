"""The GPU side of LAVAgent.run_step (team_code_v2/lav_agent_fast.py:233-323) as one object:

    half-sweep concat + ego-box removal -> ERFNet + softmax -> point painting -> 15-frame history ->
    temporal stacking in the current ego frame -> InferModel.forward -> brake net

Used by lav_amd.lav_agent.LAVAgent and by bench.py (the "full agent forward" of BASELINE.json).  Pose
bookkeeping (EKF loc/ori per frame) is handed in by the caller; nothing here touches the host except the
detection decode inside InferModel.det_inference.
"""
from __future__ import annotations

import math
import os
import time
from collections import deque

import numpy as np
import torch

from . import ops
from .model_inference import InferModel

GAP = 5  # NUM_REPEAT + 1 (lav_agent_fast.py:32-33)
_COPY_STREAM = os.environ.get("LAV_COPY_STREAM", "0") == "1"   # the frame's device-to-host copies on a stream of their own: measured 0.10 ms SLOWER with the side streams on (round 6, profiles/r06_frame_experiments.txt); off
_DET_REPORT = os.environ.get("LAV_DET_REPORT", "1") != "0"   # round 6: lav_det_decode writes the peak rows + count into pinned host memory itself and bumps a sequence word the host polls (no device->host copies, no event between the heads and the others graph)
_POSE_BLOCK = os.environ.get("LAV_POSE_BLOCK", "1") != "0"   # round 6: the 176-byte pose block rides in the staging launch's kernel arguments (no upload of its own in front of the lidar graph)
_BRAKE_AFTER = os.environ.get("LAV_BRAKE_AFTER", "in")       # "in": the brake graph starts with the frame (beside ERFNet + backbone); "feat": behind the lidar graph (beside heads / others / ego)
_BRAKE_SPLIT = os.environ.get("LAV_BRAKE_SPLIT", "wide")    # "wide" (round 6 default) | "tele": the brake net as two graphs on its stream - that image's trunk with the frame, the other's + poolings + classifier behind the lidar graph; "0": one graph with the frame
if _BRAKE_SPLIT in ("0", "off", "none"):
    _BRAKE_SPLIT = ""
_DIAG_SKIP = set(filter(None, os.environ.get("LAV_DIAG_SKIP", "").split(",")))   # timing diagnosis: skip side graphs


def ego_box_mask(lidar: torch.Tensor) -> torch.Tensor:
    """True for points inside the ego-vehicle box (lav_agent_fast.py:450-452)."""
    x, y, z = lidar[:, 0], lidar[:, 1], lidar[:, 2]
    return (x > -2.4) & (x < 0) & (y > -0.8) & (y < 0.8) & (z > -1.5) & (z < -1)


def move_lidar_points(xyz: torch.Tensor, dloc, ori0: float, ori1: float) -> torch.Tensor:
    """Re-register a past sweep into the current ego frame (lav_agent_fast.py:547-565)."""
    dloc = np.asarray(dloc, np.float64) @ np.array([[math.cos(ori0), -math.sin(ori0)], [math.sin(ori0), math.cos(ori0)]])
    ori = ori1 - ori0
    # xyz @ R + t written out (R[2] = R[:, 2] = e_z): products rounded separately and summed left to right, the
    # arithmetic lav_stack_sweeps uses - a BLAS matmul leaves summation order and FMA use unspecified
    c, s_ = float(np.float32(math.cos(ori))), float(np.float32(math.sin(ori)))
    x, y = xyz[:, 0], xyz[:, 1]
    return torch.stack([(x * c + y * -s_) + float(np.float32(dloc[0])), (x * s_ + y * c) + float(np.float32(dloc[1])),
                        xyz[:, 2]], dim=1)


def _at_frame_precision(fn):
    """Run a pipeline method with the pipeline's inference precision in force (ops.precision): the eval engines it touches are the
    ones packed for that precision - cached per precision on their modules, nothing is written onto a model somebody else owns."""
    import functools

    @functools.wraps(fn)
    def wrapped(self, *a, **k):
        with ops.precision(self.precision):
            return fn(self, *a, **k)
    return wrapped


class FramePipeline:
    def __init__(self, lidar_model, uniplanner, seg_model, bra_model, camera_x=1.5, camera_z=2.4, num_frame_stack=2,
                 device=torch.device("cuda"), compact_ego_box: bool = False):
        self.device = device
        # Round 5 put the frame's head convolution on LAV_CONV_F16X3 (two fp16 pieces per operand, three products instead of six,
        # conv_split.hpp) by writing a flag onto the LiDARModel; round 6 runs EVERY split-kernel layer of the frame that way (the scale
        # handed from layer to layer, lav_conv2d_amax) and asks for it through ops.precision around the pipeline's own calls: the
        # engines are cached per precision on their modules, a model that a trainer owns keeps its default engines.
        # LAV_INFER_PRECISION=bf16x6 (or round 5's LAV_HEADS_PRECISION=bf16x6) switches the mode off; LAV_CONV_PRECISION=f32 too.
        self.precision = ops.frame_precision()
        self.infer_model = InferModel(lidar_model, uniplanner, camera_x, camera_z, device=device, precision=self.precision)
        self.seg_model, self.bra_model = seg_model, bra_model
        self.num_frame_stack = num_frame_stack
        self.num_frame_keep = (num_frame_stack + 1) * GAP
        self.compact_ego_box = compact_ego_box
        self.reset()

    def reset(self):
        self.lidars, self.locs, self.oris = deque(), deque(), deque()
        self.prev_lidar = None

    def _sem_probs(self, all_rgbs):
        """Class probabilities of the three cameras (lav_agent_fast.py:264); lav_amd's segmenter has the softmax in its
        output layer's epilogue, any other module gets torch's."""
        probs = getattr(self.seg_model, "probs", None)
        return probs(all_rgbs) if probs is not None else torch.softmax(self.seg_model(all_rgbs), dim=1)

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
    @_at_frame_precision
    def step(self, lidar, all_rgbs, rgbs, tel_rgbs, loc, ori, nxps, cmd_value):
        """One frame.  lidar (n,4) f32, all_rgbs (3,3,288,256) f32, rgbs (1,3,288,768) f32, tel_rgbs (1,3,192,480)
        f32 - all resident in HBM; loc (2,) / ori host floats (EKF pose); nxps (2,) HBM; cmd_value int."""
        if self.prev_lidar is None:                      # first frame: only stash (lav_agent_fast.py:235-237)
            self.prev_lidar = lidar
            return None
        cur_lidar = self.preprocess(torch.cat([lidar, self.prev_lidar]))
        self.prev_lidar = lidar
        pred_sem = self._sem_probs(all_rgbs)
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


class GraphedFramePipeline(FramePipeline):
    """Same frame as FramePipeline.step, replayed from HIP graphs (torch.cuda.graphs captures the torch ops and the
    liblav_amd launches alike: both are enqueued on the capturing stream).

    Five linear graphs on three HIP streams instead of one graph with internal branches (hipGraph replays sibling
    branches of ONE graph poorly on this stack: every cross-branch edge becomes a signal wait and branches start in
    capture order; separate in-order streams overlap the way hardware queues do):

      main   lidar      merge ticks, ERFNet + softmax, painting, history + 3-sweep stacking, pillar scatter, BEV backbone
      main   heads      fused heads, peak extraction into a fixed (2,15,7) tensor
      s_bra  brake      the brake net (needs only the camera images)
      s_ego  ego[cmd]   ego crop -> ResNet-18 -> cast -> plan; starts when the feature map exists (one per command)
      main   others     a FIXED-CAPACITY batch of 15 rotated crops -> ResNet-18 -> cast -> command scores whose live row count
                        N stays in HBM (lav_det_decode at the end of the heads graph, lav_batch_limit around the branch): the
                        GPU never waits for the host between the heads and the others branch

    host:  the peak tensor and N come up behind the heads graph; the host waits for THEM only (the GPU is already in the
           others branch) and rebuilds the detection lists the API returns with the reference's score/size/range filters
           (model_inference.py:95-144).
           (`device_others=False` restores round 1's flow: copy the peaks up after the heads, decode on the host, replay
           one of 15 per-count graphs.)

    Static shapes: every LiDAR tick is padded to `points_per_tick` rows with NaN (NaN fails the pillar range test
    exactly like an absent point; all downstream results are independent of point count and order), the ego-box
    points are NaN-marked instead of compacted, and missing history sweeps are NaN slots of the ring.
    """

    def __init__(self, *a, points_per_tick: int = 32768, device_others: bool = True, **k):
        super().__init__(*a, **k)
        self.device_others = device_others and os.environ.get("LAV_DEVICE_OTHERS", "1") != "0"
        dev, P = self.device, points_per_tick
        self.P = P
        f = dict(dtype=torch.float32, device=dev)
        self.b_tick = torch.full((P, 4), float("nan"), **f)
        self.b_prev = torch.full((P, 4), float("nan"), **f)
        self.b_all_rgbs = torch.zeros((3, 3, 288, 256), **f)
        self.b_rgbs = torch.zeros((1, 3, 288, 768), **f)
        self.b_tel = torch.zeros((1, 3, 192, 480), **f)
        self.b_nxp = torch.zeros((2,), **f)
        self.ring = torch.full((self.num_frame_keep, 2 * P, 8), float("nan"), **f)
        # per-frame host values travel in ONE pinned staging buffer each way (pageable copies cost ~10 us apiece on the
        # critical path): [slot i64 | sweeps 3 x i64 | R 3x3x3 f32 | t 3x3 f32] down, the peak rows up, (loc, ori) rows down
        num_sweeps = self.num_frame_stack + 1
        if num_sweeps != 3:
            raise NotImplementedError("GraphedFramePipeline is laid out for 3 stacked sweeps (num_frame_stack = 2)")
        self.h_pose = torch.zeros((176,), dtype=torch.uint8).pin_memory()
        self.d_pose = torch.zeros((176,), dtype=torch.uint8, device=dev)
        self.b_slot, self.b_sweeps = self.d_pose[0:8].view(torch.long), self.d_pose[8:32].view(torch.long)
        self.b_R, self.b_t = self.d_pose[32:140].view(torch.float32).view(3, 3, 3), self.d_pose[140:176].view(torch.float32).view(3, 3)
        hp = self.h_pose.numpy()
        self.hn_slot, self.hn_sweeps = hp[0:8].view(np.int64), hp[8:32].view(np.int64)
        self.hn_R, self.hn_t = hp[32:140].view(np.float32).reshape(3, 3, 3), hp[140:176].view(np.float32).reshape(3, 3)
        self.h_det = torch.zeros((2, 15, 7), dtype=torch.float32).pin_memory()
        self.h_actors = torch.zeros((45,), dtype=torch.float32).pin_memory()      # [15 x (x, y) | 15 x ori]
        self.d_actors = torch.zeros((45,), **f)
        self.hn_det, self.hn_actors = self.h_det.numpy(), self.h_actors.numpy()
        self.d_n = torch.zeros((1,), dtype=torch.int32, device=dev)               # number of other vehicles, device resident
        self.h_n = torch.zeros((1,), dtype=torch.int32).pin_memory()
        self.h_seq = torch.zeros((1,), dtype=torch.int32).pin_memory()             # bumped by lav_det_decode_report once rows + count are on the host
        self.hn_seq = self.h_seq.numpy()
        self.ev_det, self.ev_end = torch.cuda.Event(), torch.cuda.Event()
        self.b_features = None   # (1,384,160,160): written by the lidar graph, read by heads / ego / others
        self.b_locs, self.b_oris = self.d_actors[:30].view(15, 2), self.d_actors[30:]
        self.b_zero = torch.zeros((1, 4), **f)   # the ego vehicle's own (loc, ori)
        # capture streams double as workspace keys (lav_amd.ops._workspace): graphs that run concurrently must not
        # share split-K scratch, graphs of one stream may
        # (LAV_BRAKE_PRIORITY / LAV_EGO_PRIORITY: HIP stream priority of the side streams, e.g. 1 = low; default 0.  tools/frame_ab.py)
        self.s_cap = torch.cuda.Stream(dev)
        self.s_bra = torch.cuda.Stream(dev, priority=int(os.environ.get("LAV_BRAKE_PRIORITY", "0")))
        self.s_ego = torch.cuda.Stream(dev, priority=int(os.environ.get("LAV_EGO_PRIORITY", "0")))
        self.ev_in, self.ev_feat, self.ev_heads = torch.cuda.Event(), torch.cuda.Event(), torch.cuda.Event()
        self.s_copy = torch.cuda.Stream(dev)   # the frame's device-to-host copies (peak rows, vehicle count)
        # the brake net runs beside the LiDAR chain: its layers are planned for part of the chip so that both fit
        side = int(os.environ.get("LAV_BRAKE_CUS", "128"))
        trunk = getattr(self.bra_model, "conv_backbone", None)
        if side and trunk is not None:
            object.__setattr__(trunk, "target_cus", side)
            trunk._drop()
        self.graphs, self.outs = {}, {}
        self.frame_no = 0
        self._h2d_stage = {}          # device staging tensors of host inputs, by (input, shape, strides)
        self.host_trace = [] if os.environ.get("LAV_FRAME_HOST_TRACE") else None   # (label, perf_counter) stamps of step()
        self.overflow_ticks = 0       # ticks larger than the static buffers (each grew them and re-captured the graphs)
        self.forced_others = None     # set_forced_others: (actors, count, n) on the device
        self.decode_mismatches = 0    # frames on which the device and host detection decodes disagreed (host result used)
        self.plan_aborts = 0          # frames whose persistent plan launch timed out and was recomputed (recover_plan)
        self.nonfinite_det_frames = 0 # frames whose peak rows (scores, positions, sizes, orientations) held a NaN / Inf (host check)
        # [tensors found non-finite, checks run] by lav_nonfinite_count at the end of the ego / brake graphs: sticky and
        # device resident, read by health() - a benchmark loop proves every timed frame finite without a copy per frame
        self.d_health = torch.zeros((2,), dtype=torch.int32, device=dev)
        self.poses = deque()

    def _grow(self, new_p: int):
        """Re-allocate the per-tick static buffers for `new_p` points per tick, keeping the history, and forget the graphs."""
        torch.cuda.synchronize()
        f = dict(dtype=torch.float32, device=self.device)
        old_p = self.P
        b_tick = torch.full((new_p, 4), float("nan"), **f)
        b_prev = torch.full((new_p, 4), float("nan"), **f); b_prev[:old_p] = self.b_prev
        ring = torch.full((self.num_frame_keep, 2 * new_p, 8), float("nan"), **f); ring[:, :2 * old_p] = self.ring
        self.b_tick, self.b_prev, self.ring, self.P = b_tick, b_prev, ring, new_p
        self.graphs.clear(); self.outs.clear()
        torch.cuda.synchronize()

    def reset(self):
        super().reset()
        if hasattr(self, "ring"):
            self.ring.fill_(float("nan")); self.b_prev.fill_(float("nan"))
            self.frame_no = 0
            self.poses = deque()

    # ---- the graphs' bodies --------------------------------------------------------------------------------
    def _g_lidar(self):
        im, lm = self.infer_model, self.infer_model.lidar_model
        cur = ops.merge_ticks(self.b_tick, self.b_prev)                 # concat + ego box + prev <- tick: one launch
        pred_sem = self._sem_probs(self.b_all_rgbs)
        fused = im.forward_paint(cur, pred_sem)
        lidar_points = ops.stack_sweeps(fused, self.ring, self.b_slot, self.b_sweeps, self.b_R, self.b_t)
        canvas = lm.point_pillar_net([lidar_points], [lidar_points.shape[0]])
        if self.b_features is None:
            self.b_features = torch.empty((1, lm.backbone.out_channels, canvas.shape[2] // 2, canvas.shape[3] // 2),
                                          dtype=torch.float32, device=canvas.device)
        lm.backbone(canvas, out=self.b_features)
        return dict(lidar_points=lidar_points)

    def _g_heads(self):
        heat, size, ori, pred_bev = self.infer_model.lidar_model.heads(self.b_features)
        det_raw = ops.extract_peaks(heat[0], size[0], ori[0], apply_sigmoid=True)                                 # (2,15,7)
        if self.device_others:   # who the other vehicles are and how many: decided on the device (model_inference.py:95-144)
            im, up = self.infer_model, self.infer_model.uniplanner
            ox, oy = up.offsets()
            H, W = im._bev_hw
            ops.det_decode(det_raw, self.d_actors, self.d_n, cls=1, min_score=0.2, ego_xy=(160, 280), near_px=2.0,
                           far_px=30 * im.pixels_per_meter, min_box=0.1 * im.pixels_per_meter,
                           centre_xy=(float(W / 2 + ox * W / 2), float(H / 2 + oy * H / 2)), skip_px=4.0, ppm=up.pixels_per_meter,
                           report=(self.h_det, self.h_n, self.h_seq) if _DET_REPORT else None)
            if self.forced_others is not None:   # measurement hook (set_forced_others): fixed poses instead of the detections
                self.d_actors.copy_(self.forced_others[0]); self.d_n.copy_(self.forced_others[1])
        return dict(det_raw=det_raw, pred_bev=pred_bev)

    def set_forced_others(self, locs=None, oris=None):
        """Measurement hook ("benchmark with the detection list forced to N in {0, 4} fixed poses"): from the next frame
        on the others branch runs on these ego-frame poses (locs (n,2) metres, oris (n,) rad; n <= 15) instead of the decoded
        detections - random weights detect an arbitrary number of vehicles, so frame times are only comparable across weights
        at a fixed n.  None restores the detections.  The returned `det` lists stay the decoded ones."""
        if not self.device_others:
            raise RuntimeError("set_forced_others needs the device-resident others branch")
        if locs is None:
            self.forced_others = None
        else:
            n = len(locs)
            host = torch.zeros((45,), dtype=torch.float32)
            host[:2 * n] = torch.as_tensor(locs, dtype=torch.float32).reshape(-1)
            host[30:30 + n] = torch.as_tensor(oris, dtype=torch.float32).reshape(-1)
            self.forced_others = (host.to(self.device), torch.tensor([n], dtype=torch.int32, device=self.device), n)
        torch.cuda.synchronize()
        self.graphs.pop("heads", None); self.outs.pop("heads", None)   # the hook is part of the heads graph: re-capture

    def _g_brake(self):
        pred_bra = self.bra_model(self.b_rgbs, self.b_tel)
        ops.nonfinite_count([pred_bra], self.d_health)
        return dict(pred_bra=pred_bra)

    # LAV_BRAKE_SPLIT (round 6): the brake net as two graphs on its stream - the first image's trunk with the frame, the second image's trunk +
    # poolings + classifier behind the lidar graph ("wide", the default: wide image first; "tele": tele image first).  The brake net costs the
    # frame through the lidar graph it runs beside (0.18 ms): half of it leaves that prefix, the suffix - where the result is not needed before
    # the frame's end - takes it: 1.969-1.982 vs 1.983-1.989 ms in three interleaved pairs (profiles/r06_frame_experiments.txt section 10)
    def _g_brake_a(self):
        first = self.b_rgbs if _BRAKE_SPLIT == "wide" else self.b_tel
        return dict(x=self.bra_model.trunk(first))

    def _g_brake_b(self):
        xa = self.outs["brake_a"]["x"]
        xb = self.bra_model.trunk(self.b_tel if _BRAKE_SPLIT == "wide" else self.b_rgbs)
        x1, x2 = (xa, xb) if _BRAKE_SPLIT == "wide" else (xb, xa)
        pred_bra = self.bra_model.classify(x1, x2)
        ops.nonfinite_count([pred_bra], self.d_health)
        return dict(pred_bra=pred_bra)

    def _g_ego(self, cmd_value):
        up, features = self.infer_model.uniplanner, self.b_features
        ego_crop = up.crop_feature(features, self.b_zero[:, :2], self.b_zero[0, :1], up.pixels_per_meter / 2, up.crop_size, amax=ops.amax_of(features))
        ego_embd, ego_cast, _ = up.embed_cast(ego_crop)
        ego_plan = up.plan(ego_embd, self.b_nxp[None], cast_locs=ego_cast, pixels_per_meter=up.pixels_per_meter,
                           crop_size=up.crop_size * 2, cmd=int(cmd_value))[0, -1, 0]
        ops.nonfinite_count([ego_embd, ego_plan, ego_cast], self.d_health)
        return dict(ego_embd=ego_embd, ego_plan_locs=ego_plan, ego_cast_locs=ego_cast[0, int(cmd_value)])

    def _g_others(self, n):
        up = self.infer_model.uniplanner
        feats = self.b_features
        locs, oris = self.b_locs[:n], self.b_oris[:n]
        crops = up.crop_feature(feats.expand(n, -1, -1, -1), locs, oris, up.pixels_per_meter / 2, up.crop_size, amax=ops.amax_of(feats))
        _, cast, cmds = up.embed_cast(crops, oris=oris, locs=locs, want_cmds=True)   # pool, cast GRUs, command scores, ego frame
        return dict(other_cast_locs=cast, other_cast_cmds=cmds)

    def _g_others_cap(self):
        """The others branch at its full capacity of 15 vehicles; kernels skip the rows >= d_n (read on the device)."""
        with ops.batch_limit(self.d_n):
            return self._g_others(15)

    def _stamp(self, label):
        """Host-side time stamps of one step (LAV_FRAME_HOST_TRACE=1, tools/host_tail_probe.py): where the host spends the frame."""
        if self.host_trace is not None:
            self.host_trace.append((label, time.perf_counter()))

    def _replay(self, key, fn, stream, *args, _skip=False):
        """Replay graph `key` on the current stream; first use: one eager run on the capture stream (builds the
        convolution engines and the per-stream workspaces), then the capture.  Every graph has a PRIVATE memory pool:
        graphs that share one may only be replayed in capture order."""
        g = self.graphs.get(key)
        if g is None:
            torch.cuda.synchronize()
            state = (self.ring.clone(), self.b_prev.clone())   # the lidar graph advances these: undo the two extra runs
            with torch.cuda.stream(stream):
                fn(*args)
            torch.cuda.synchronize()
            self.ring.copy_(state[0]); self.b_prev.copy_(state[1])
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=stream):
                out = fn(*args)
            torch.cuda.synchronize()
            self.ring.copy_(state[0]); self.b_prev.copy_(state[1])
            self.graphs[key], self.outs[key] = g, out
        if not _skip:     # _skip: diagnosis only (LAV_DIAG_SKIP), outputs keep their last values
            g.replay()
        return self.outs[key]

    def _set_pose_buffers(self, upload=True):
        """History indices and the float32 rotation / translation of every stacked sweep (lav_agent_fast.py:363-383,
        547-565), computed on the host exactly like the reference and uploaded as three tiny tensors."""
        n_hist = min(self.frame_no + 1, self.num_frame_keep)
        slot = self.frame_no % self.num_frame_keep
        loc0, ori0 = self.poses[-1]
        sweeps, Rs, ts = [], [], []
        for i in range(self.num_frame_stack + 1):
            age = i * GAP
            if age < n_hist:
                loc, ori = self.poses[-1 - age]
                dloc = (np.asarray(loc, np.float64) - loc0) @ np.array([[math.cos(ori0), -math.sin(ori0)], [math.sin(ori0), math.cos(ori0)]])
                o = ori - ori0
                Rs.append([[math.cos(o), math.sin(o), 0.], [-math.sin(o), math.cos(o), 0.], [0., 0., 1.]])
                ts.append([float(dloc[0]), float(dloc[1]), 0.])
                sweeps.append((slot - age) % self.num_frame_keep)
            else:  # history not that deep yet: point at a slot that is still all-NaN (never written) or stale-free
                Rs.append([[1., 0., 0.], [0., 1., 0.], [0., 0., 1.]]); ts.append([0., 0., 0.])
                sweeps.append((slot - age) % self.num_frame_keep)
        self.hn_slot[0] = slot
        self.hn_sweeps[:] = sweeps
        self.hn_R[:] = np.asarray(Rs, np.float64).astype(np.float32)
        self.hn_t[:] = np.asarray(ts, np.float64).astype(np.float32)
        if upload:
            self.d_pose.copy_(self.h_pose, non_blocking=True)

    @torch.no_grad()
    @_at_frame_precision
    def step(self, lidar, all_rgbs, rgbs, tel_rgbs, loc, ori, nxps, cmd_value):
        n = int(lidar.shape[0])
        if n > self.P:
            # The reference takes any tick size; the graphs' buffers are static.  A tick that does not fit GROWS them (tick and
            # previous-tick buffers, the 15-slot history ring - old rows kept, the new ones NaN = absent points) and drops the
            # captured graphs, which are re-captured on this frame: exact results, one slow frame (a warning says so).  Construct
            # the pipeline with points_per_tick >= the sensor's points_per_second / 20 / 2 to never get here.
            import warnings
            new_p = (n + 8191) // 8192 * 8192
            warnings.warn(f"LiDAR tick of {n} points exceeds the static graph buffers (points_per_tick={self.P}): "
                          f"growing them to {new_p} and re-capturing the frame graphs")
            self.overflow_ticks += 1
            self._grow(new_p)
        if self.prev_lidar is None:                      # first frame only stashes the tick (lav_agent_fast.py:235-237)
            self.b_tick[:n].copy_(lidar[:n], non_blocking=True)
            if n < self.P:
                self.b_tick[n:].fill_(float("nan"))
            self.b_prev.copy_(self.b_tick)
            self.prev_lidar = True
            return None
        self._stamp("enter")
        # the tick's LiDAR rows, camera tensors and next waypoint into the graphs' static buffers: one launch (tensors that are
        # not float32 / contiguous / device resident take Tensor.copy_ inside copy_many)
        pairs = [(self.b_tick[:n], lidar[:n]), (self.b_all_rgbs, all_rgbs), (self.b_rgbs, rgbs), (self.b_tel, tel_rgbs), (self.b_nxp, nxps)]
        # HOST tensors (the sensor path of a real drive) travel as they lie in memory - strides kept - into device staging tensors
        # (asynchronously when they are pinned), and the layout change happens on the GPU with the rest of copy_many.  Copied straight
        # into the static buffers, a channels-last camera tensor is permuted by torch ON THE CPU first: 37 ms of host time per frame
        # (tools/upload_probe.py: 39.8 frames/s against 402 this way and 431 with resident inputs).
        for k, (b, t) in enumerate(pairs):
            if isinstance(t, torch.Tensor) and not t.is_cuda and t.dtype == torch.float32:
                key = (k, tuple(t.shape), tuple(t.stride()))
                st = self._h2d_stage.get(key)
                if st is None:
                    if len(self._h2d_stage) > 64:
                        self._h2d_stage.clear()
                    st = self._h2d_stage[key] = torch.empty_like(t, device=self.device)   # (preserve_format: dense layouts keep their strides)
                st.copy_(t, non_blocking=t.is_pinned())
                pairs[k] = (b, st)
        self.poses.append((np.asarray(loc, np.float64), float(ori)))
        if len(self.poses) > self.num_frame_keep:
            self.poses.popleft()
        if all(isinstance(t, torch.Tensor) and t.is_cuda and t.shape == b.shape for b, t in pairs):
            self._set_pose_buffers(upload=not _POSE_BLOCK)
            ops.copy_many(pairs, block=(self.d_pose, self.h_pose) if _POSE_BLOCK else None)
        else:
            for b, t in pairs:
                b.copy_(t, non_blocking=True)
            self._set_pose_buffers()
        if n < self.P:
            self.b_tick[n:].fill_(float("nan"))
        cmd_value = int(cmd_value)
        main = torch.cuda.current_stream()
        brake_late = _BRAKE_AFTER == "feat"
        if not brake_late:
            self.ev_in.record(main)                                    # inputs are in their static buffers
        self._stamp("inputs enqueued")
        o_lidar = self._replay("lidar", self._g_lidar, self.s_cap)
        self._stamp("lidar graph launched")
        if not brake_late:
            self.s_bra.wait_event(self.ev_in)
            with torch.cuda.stream(self.s_bra):
                if _BRAKE_SPLIT:
                    self._replay("brake_a", self._g_brake_a, self.s_bra, _skip="brake" in _DIAG_SKIP and "brake_a" in self.graphs)
                else:
                    o_bra = self._replay("brake", self._g_brake, self.s_bra, _skip="brake" in _DIAG_SKIP and "brake" in self.graphs)
        self.ev_feat.record(main)                                      # feature map complete
        if _BRAKE_SPLIT and not brake_late:
            skip_b = "brake" in _DIAG_SKIP and "brake_b" in self.graphs
            if not skip_b:
                self.s_bra.wait_event(self.ev_feat)
            with torch.cuda.stream(self.s_bra):
                o_bra = self._replay("brake_b", self._g_brake_b, self.s_bra, _skip=skip_b)
        if brake_late:
            self.s_bra.wait_event(self.ev_feat)
            with torch.cuda.stream(self.s_bra):
                o_bra = self._replay("brake", self._g_brake, self.s_bra, _skip="brake" in _DIAG_SKIP and "brake" in self.graphs)
        self.s_ego.wait_event(self.ev_feat)
        with torch.cuda.stream(self.s_ego):
            o_ego = self._replay(("ego", cmd_value), self._g_ego, self.s_ego, cmd_value,
                                 _skip="ego" in _DIAG_SKIP and ("ego", cmd_value) in self.graphs)
        seq0 = int(self.hn_seq[0])
        o_heads = self._replay("heads", self._g_heads, self.s_cap)
        self._stamp("brake, ego, heads graphs launched")
        self.frame_no += 1
        up = self.infer_model.uniplanner
        if self.device_others:
            # peaks and count travel up while the GPU goes straight on with the others branch; the host reads them (and rebuilds
            # the detection lists of the API) while that branch runs - nothing on the GPU waits for it
            # (round 6: on a copy stream of their own - issued on the main stream the two copies sat between the heads graph and the
            # others graph, 25 us of the critical chain in the kernel trace)
            if _COPY_STREAM:
                self.ev_heads.record(main)
                self.s_copy.wait_event(self.ev_heads)
                with torch.cuda.stream(self.s_copy):
                    self.h_det.copy_(o_heads["det_raw"], non_blocking=True)
                    self.h_n.copy_(self.d_n, non_blocking=True)
                    self.ev_det.record(self.s_copy)
            elif not _DET_REPORT:
                self.h_det.copy_(o_heads["det_raw"], non_blocking=True)
                self.h_n.copy_(self.d_n, non_blocking=True)
                self.ev_det.record(main)
            ob = self._replay("others_cap", self._g_others_cap, self.s_cap)
            main.wait_stream(self.s_ego)      # also keeps the next frame's input copies behind this frame's readers
            main.wait_stream(self.s_bra)
            if _COPY_STREAM:
                main.wait_stream(self.s_copy)     # (the next frame's heads graph rewrites the rows being copied)
            self._stamp("others graph launched")
            if _DET_REPORT and not _COPY_STREAM:
                self._wait_report(seq0)
            else:
                self.ev_det.synchronize()
            self._stamp("heads done on the GPU (event)")
            if not np.isfinite(self.hn_det).all():   # the peak rows are on the host every frame: their check costs no launch
                self.nonfinite_det_frames += 1
            det, locs, oris = self.infer_model.det_decode_fast(self.hn_det)
            N = int(self.h_n[0])
            if self.forced_others is not None:
                N = self.forced_others[2]
            elif N != min(len(locs), 15):
                # Both sides apply the same float64 rules to the same float32 rows (lav_det_decode takes its thresholds as
                # doubles), so this should not happen; a drive must not die on it if it does: the host-decoded actors are
                # uploaded and the branch is replayed for that count (a per-count graph, captured on first use).
                import warnings
                warnings.warn(f"device detection decode found {N} vehicles, the host rules {len(locs)}: using the host's")
                self.decode_mismatches += 1
                N = min(len(locs), 15)
                if N > 0:
                    self.hn_actors[:] = 0
                    self.hn_actors[:2 * N] = locs[:N].reshape(-1)
                    self.hn_actors[30:30 + N] = oris[:N]
                    self.d_actors.copy_(self.h_actors, non_blocking=True)
                    ob = self._replay(("others", N), self._g_others, self.s_cap, N)
            if N > 0:
                other_cast, other_cmds = ob["other_cast_locs"][:N], ob["other_cast_cmds"][:N]
            else:   # the reference returns CPU zeros (model_inference.py:167-168)
                other_cast = torch.zeros((0, up.num_cmds, up.num_plan, 2))
                other_cmds = torch.zeros((0, up.num_cmds))
            self._stamp("decoded, return")
            return dict(ego_embd=o_ego["ego_embd"], ego_plan_locs=o_ego["ego_plan_locs"], ego_cast_locs=o_ego["ego_cast_locs"],
                        other_cast_locs=other_cast, other_cast_cmds=other_cmds, pred_bev=o_heads["pred_bev"],
                        det=det, pred_bra=o_bra["pred_bra"], lidar_points=o_lidar["lidar_points"])
        # the frame's only blocking device->host copy before the others branch: 840 bytes into pinned memory
        self.h_det.copy_(o_heads["det_raw"], non_blocking=True)
        self.ev_det.record(main)
        self.ev_det.synchronize()
        det, locs, oris = self.infer_model.det_decode_fast(self.hn_det)     # numpy masks: ~3x faster than the row loops
        N = min(len(locs), 15)
        if N > 0:
            self.hn_actors[:2 * N] = locs[:N].reshape(-1)
            self.hn_actors[30:30 + N] = oris[:N]
            self.d_actors.copy_(self.h_actors, non_blocking=True)
            ob = self._replay(("others", N), self._g_others, self.s_cap, N)
            other_cast, other_cmds = ob["other_cast_locs"], ob["other_cast_cmds"]
        else:
            other_cast = torch.zeros((0, up.num_cmds, up.num_plan, 2))
            other_cmds = torch.zeros((0, up.num_cmds))
        main.wait_stream(self.s_ego)      # also keeps the next frame's input copies behind this frame's readers
        main.wait_stream(self.s_bra)
        return dict(ego_embd=o_ego["ego_embd"], ego_plan_locs=o_ego["ego_plan_locs"], ego_cast_locs=o_ego["ego_cast_locs"],
                    other_cast_locs=other_cast, other_cast_cmds=other_cmds, pred_bev=o_heads["pred_bev"],
                    det=det, pred_bra=o_bra["pred_bra"], lidar_points=o_lidar["lidar_points"])

    def _wait_report(self, seq0, timeout_s=10.0):
        """Spin until lav_det_decode_report has bumped the pinned sequence word past `seq0`: the peak rows and the count it wrote before
        the word are then in h_det / h_n (system-scope release on the device, loads in program order here)."""
        hs, spins, t_end = self.hn_seq, 0, None
        while int(hs[0]) == seq0:
            spins += 1
            if spins & 0x3ff == 0:
                now = time.perf_counter()
                if t_end is None:
                    t_end = now + timeout_s
                elif now > t_end:
                    torch.cuda.synchronize()
                    if int(hs[0]) == seq0:
                        raise RuntimeError("lav_det_decode_report: the heads graph completed without reporting its peak rows to the host")

    @torch.no_grad()
    @_at_frame_precision
    def recover_plan(self, out, cmd_value):
        """Called by the consumer of step()'s result when `ego_plan_locs` holds NaN: if the persistent plan kernel gave up
        (its 64 workgroups were not co-resident beside the other streams' work - status word of its workspace), warn and
        recompute this frame's plan on the step-per-launch path, which has no such requirement.  Returns the waypoints."""
        up = self.infer_model.uniplanner
        status = ops.gru_plan_status(1, up.plan_gru.hidden_size, up.num_cmds, int(cmd_value), self.device, stream=self.s_ego)
        if status == 0:
            return out["ego_plan_locs"]          # NaN from the network itself: the reference's rule for it applies
        import warnings
        warnings.warn("lav_gru_plan: the persistent plan kernel timed out (GPU oversubscribed); recomputing on the step path")
        ego_cast = up.cast(out["ego_embd"], mode="ego")
        plan = up.plan(out["ego_embd"], self.b_nxp[None], cast_locs=ego_cast, pixels_per_meter=up.pixels_per_meter,
                       crop_size=up.crop_size * 2, cmd=int(cmd_value), impl="steps")[0, -1, 0]
        self.plan_aborts += 1
        return plan

    @torch.no_grad()
    @_at_frame_precision
    def plan_deviation(self, out, cmd_value) -> float:
        """max |graph's persistent plan - the same plan recomputed on the step-per-launch path| of a frame `step` returned (the two
        paths are bit-identical: tests/test_gpu_paint_gru.py).  A finite but wrong plan is invisible to the health counters - round
        4's quarter-poll kernel produced exactly that beside the crop stems - so bench.py runs this on frames after its timed ones."""
        up = self.infer_model.uniplanner
        with torch.cuda.stream(self.s_ego):
            self.s_ego.wait_stream(torch.cuda.current_stream())
            ego_cast = up.cast(out["ego_embd"], mode="ego")
            plan = up.plan(out["ego_embd"], self.b_nxp[None], cast_locs=ego_cast, pixels_per_meter=up.pixels_per_meter,
                           crop_size=up.crop_size * 2, cmd=int(cmd_value), impl="steps")[0, -1, 0]
            dev = (plan - out["ego_plan_locs"]).abs().max()
        torch.cuda.current_stream().wait_stream(self.s_ego)
        return float(dev)

    def health(self, cmd_value: int = 3) -> dict:
        """Counters of the drive so far (synchronises): non-finite output tensors seen by the graphs' own checks (ego embedding /
        plan / cast, brake prediction; the peak rows are checked on the host), persistent plan launches and how many of them timed out (sticky words of the
        plan workspace on the ego stream - they count what the GPU did, whether or not the caller looked at the waypoints),
        plans recomputed by recover_plan, device / host detection-decode disagreements, ticks that outgrew the static buffers."""
        torch.cuda.synchronize()
        up = self.infer_model.uniplanner
        diag = ops.gru_plan_diag(1, up.plan_gru.hidden_size, up.num_cmds, int(cmd_value), self.device, stream=self.s_ego)
        nonfinite, checks = (int(v) for v in self.d_health.cpu())
        chain_timeouts, chain_launches = ops.pair_chain_status(self.device, stream=self.s_cap)   # ERFNet's persistent pair runs (lidar graph)
        nonfinite += self.nonfinite_det_frames
        return dict(nonfinite_outputs=nonfinite, finite_checks=checks, plan_launches=diag["launches"], plan_aborts=diag["aborted_launches"],
                    pair_chain_launches=chain_launches, pair_chain_timeouts=chain_timeouts,
                    plans_recomputed=self.plan_aborts, decode_mismatches=self.decode_mismatches, overflow_ticks=self.overflow_ticks,
                    last_plan_launch=diag)

    @torch.no_grad()
    @_at_frame_precision
    def precapture(self, cmds=range(6), max_others=15):
        """Capture every graph a drive will need up front - the frame graphs, one ego graph per command value, one others
        graph per vehicle count up to `max_others` - so that no 20 Hz tick pays a capture (hundreds of ms) the first time a
        command or a vehicle count occurs.  Runs two synthetic ticks and then resets the pipeline."""
        dev, P = self.device, self.P
        tick = torch.zeros((P, 4), dtype=torch.float32, device=dev)
        tick[:, 0] = torch.linspace(5.0, 60.0, P, device=dev); tick[:, 2] = -1.0
        rgb = lambda *shape: torch.zeros(shape, dtype=torch.float32, device=dev)
        args = (rgb(3, 3, 288, 256), rgb(1, 3, 288, 768), rgb(1, 3, 192, 480), np.zeros(2), 0.0, torch.zeros(2, device=dev))
        self.step(tick, *args, 3)
        self.step(tick, *args, 3)
        for cmd in cmds:
            with torch.cuda.stream(self.s_ego):
                self._replay(("ego", int(cmd)), self._g_ego, self.s_ego, int(cmd))
        if self.device_others:
            self._replay("others_cap", self._g_others_cap, self.s_cap)
        else:
            for n in range(1, max_others + 1):
                self._replay(("others", n), self._g_others, self.s_cap, n)
        torch.cuda.synchronize()
        self.reset()

    def _decode(self, det_rows, min_score=0.2):
        return self.infer_model.det_decode(det_rows, min_score)
