#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REFERENCE's own
Python modules on CPU.

Runs only in the build container (needs /root/reference, read-only).  The reference
cannot travel to the GPU box, so the outputs are committed as small .npz fixtures and
this script is committed with them.  Inputs and weights are NOT stored: they are
regenerated from lav_amd.synth (numpy PCG64 keyed by (seed, name)); each fixture
carries CRC32s of its inputs so a drifted generator is detected, not silently compared.

Stand-ins for the two packages the reference imports but this image lacks live in
tests/golden/_shims (torch_scatter, carla) - see their docstrings.

    python tests/golden/make_golden.py            # rewrites tests/golden/*.npz
"""
import os
import sys
import zlib

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("LAV_REFERENCE", "/root/reference")
sys.path[:0] = [os.path.join(HERE, "_shims"), os.path.join(REF, "team_code_v2"), REPO]

import numpy as np  # noqa: E402
import torch  # noqa: E402

from models.lidar import LiDARModel  # noqa: E402  (reference)
from models.uniplanner import UniPlanner  # noqa: E402  (reference)
from models.bev_planner import BEVPlanner  # noqa: E402  (reference)
from models.rgb import RGBSegmentationModel, RGBBrakePredictionModel  # noqa: E402  (reference)
import model_inference as ref_mi  # noqa: E402  (reference)

from lav_amd import synth  # noqa: E402

torch.set_grad_enabled(False)
torch.manual_seed(0)

CFG = dict(min_x=-10, max_x=70, min_y=-40, max_y=40, pixels_per_meter=4)


def crc(a) -> int:
    return zlib.crc32(np.ascontiguousarray(a).tobytes())


def build_reference():
    lm = LiDARModel(num_input=16, backbone="cnn", num_features=[64, 64], **CFG)
    y_off = 1 + CFG["min_x"] / ((CFG["max_x"] - CFG["min_x"]) / 2)
    bp = BEVPlanner(pixels_per_meter=4, crop_size=96, feature_x_jitter=1.5, feature_angle_jitter=20,
                    x_offset=0, y_offset=y_off, num_cmds=6, num_plan=20, num_plan_iter=5, num_frame_stack=2)
    up = UniPlanner(bp, pixels_per_meter=4, crop_size=96, feature_x_jitter=1.5, feature_angle_jitter=20,
                    x_offset=0, y_offset=y_off, num_cmds=6, num_plan=20, num_input_feature=384, num_plan_iter=5)
    lm.load_state_dict(synth.seeded_state_dict(lm, prefix="lidar."))
    up.load_state_dict(synth.seeded_state_dict(up, prefix="uni."))
    return lm.eval(), up.eval()


def save(name, **arrs):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrs.items()})
    print(f"{name}.npz  {os.path.getsize(path) / 1024:.0f} KiB")


def t2n(t):
    return t.detach().cpu().numpy()


# ----------------------------------------------------------------------------------------------
def edge_points():
    """Hand-made cloud hitting every boundary rule of grid_locations (point_pillar.py:70-79)."""
    e = np.nextafter
    f = np.float32
    rows = [
        [-10.0, -40.0, 0.0], [e(f(-10), f(-11)), 0.0, 0.0], [e(f(70), f(0)), e(f(40), f(0)), 1.0],
        [70.0, 0.0, 0.0], [0.0, 40.0, 0.0], [-0.0, -0.0, -1.2], [0.0, 0.0, -1.3], [0.24999, 0.25, 0.5],
        [0.25, 0.24999, 0.5], [69.99999, 39.99999, 0.1], [12.125, -7.875, 0.3], [12.125, -7.875, 0.9],
        [np.nan, 1.0, 0.0], [1.0, np.nan, 0.0], [np.inf, 0.0, 0.0], [-np.inf, 0.0, 0.0], [5.0, 5.0, 0.7],
        [-9.999999, -39.999996, 2.0], [3.3333333, -3.3333333, 0.0], [3.3333335, -3.3333335, 0.0],
    ]
    pts = np.zeros((len(rows), 11), np.float32)
    pts[:, :3] = np.asarray(rows, np.float32)
    r = np.random.Generator(np.random.PCG64(7))
    pts[:, 3:8] = r.uniform(0, 1, (len(rows), 5)).astype(np.float32)
    pts[:, 8 + (np.arange(len(rows)) % 3)] = 1
    return pts


def gold_pillar(lm):
    ppn = lm.point_pillar_net
    out = {}
    cases = {
        "lidar": [synth.stacked_lidar(512, kind="lidar")],
        "uniform": [synth.stacked_lidar(512, kind="uniform")],
        "edge": [edge_points()],
        "one_cell": [np.tile(np.array([[3.1, 4.1, -0.5, .2, 0, 0, 0, 0, 1, 0, 0]], np.float32), (70, 1))
                     + np.linspace(0, 0.1, 70, dtype=np.float32)[:, None] * np.eye(11, dtype=np.float32)[2]],
        "all_outside": [np.concatenate([synth.stacked_lidar(64)[:, :11] * 0 + 100.0])],
    }
    # training-style padded batch of 2 with num_points truncation (point_pillar.py:98)
    a = synth.stacked_lidar(300, seed=11, kind="uniform")
    b = synth.stacked_lidar(300, seed=12, kind="lidar")
    cases["batch2"] = [a, b]
    nump = {"batch2": [700, 450]}
    for name, lst in cases.items():
        n = nump.get(name, [len(p) for p in lst])
        tl = [torch.from_numpy(p) for p in lst]
        # replicate forward() stage by stage to expose the intermediate indices
        coords, pts = [], []
        for bi, p in enumerate(tl):
            p = p[: n[bi]]
            p, g = ppn.grid_locations(p)
            coords.append(torch.nn.functional.pad(g, (1, 0), value=bi))
            pts.append(p)
        coords, pts = torch.cat(coords), torch.cat(pts)
        if len(pts) > 0:
            dec, uniq, inv = ppn.pillar_generation(pts, coords)
            feat = ppn.point_net(dec, inv)
            canvas = ppn.scatter_points(feat, uniq, len(tl))
            full = ppn(tl, n)
            assert torch.equal(full, canvas)
        else:  # reference raises on an empty cloud (unique of empty / scatter of empty): record that
            uniq = torch.zeros((0, 3), dtype=torch.long); inv = torch.zeros((0,), dtype=torch.long)
            feat = torch.zeros((0, 64)); dec = torch.zeros((0, 16))
        out[f"{name}/in_crc"] = np.array([crc(p) for p in lst], np.int64)
        out[f"{name}/num_points"] = np.array(n, np.int64)
        out[f"{name}/unique_coords"] = t2n(uniq).astype(np.int32)
        out[f"{name}/inverse"] = t2n(inv).astype(np.int32)
        out[f"{name}/feat"] = t2n(feat)
        out[f"{name}/decorated"] = t2n(dec)
    out["edge/points"] = edge_points()
    save("pillar", **out)


def gold_paint(lm, up):
    im = ref_mi.InferModel(lm, up, 1.5, 2.4, device=torch.device("cpu"))
    lidar = np.concatenate([synth.lidar_sweep(2048, name="paintA"), synth.lidar_sweep(2048, name="paintB")])
    # a few hand-made points: behind every camera, on the optical axes, at the image borders, ego box
    extra = np.array([[1.5, 0, 2.4 - 2.4, .5], [1.5 + 1e-5, 0, 0, .5], [10, 0, 0, .5], [-10, 0, 0, .5],
                      [5, 8.66, 0, .1], [5, -8.66, 0, .1], [1.49999, 0, 0, .3], [30, 18.7, -2.4, .3],
                      [-1.0, 0.0, -1.2, .9], [20, 12.4967, 3, .2], [20, -12.4967, 3, .2]], np.float32)
    lidar = np.concatenate([lidar, extra]).astype(np.float32)
    sem = synth.semantic_probs()
    fused = im.forward_paint(torch.from_numpy(lidar), torch.from_numpy(sem))
    uvz = [t2n(cc(torch.from_numpy(lidar))).astype(np.int64) for cc in im.coord_converters]
    mats = {}
    for i, cc in enumerate(im.coord_converters):
        mats[f"K{i}"] = t2n(cc.K); mats[f"l2w{i}"] = t2n(cc.lidar_to_world); mats[f"w2c{i}"] = t2n(cc.world_to_cam)
    save("paint", lidar=lidar, sem_crc=np.array([crc(sem)], np.int64), fused=t2n(fused),
         uvz=np.stack(uvz).clip(-2 ** 31, 2 ** 31 - 1).astype(np.int32), **mats)


def gold_bev(lm, up):
    pts = synth.stacked_lidar(8192)
    canvas = lm.point_pillar_net([torch.from_numpy(pts)], [len(pts)])
    bb = lm.backbone
    x1 = bb.conv1(canvas); x2 = bb.conv2(x1); x3 = bb.conv3(x2)
    feat = bb(canvas)
    heads = [lm.center_head(feat), lm.box_head(feat), lm.ori_head(feat), lm.seg_head(feat)]
    save("bev", in_crc=np.array([crc(pts)], np.int64),
         canvas_sum=t2n(canvas.double().sum((2, 3))[0]),
         x1_s=t2n(x1[0, :, ::8, ::8]), x2_s=t2n(x2[0, :, ::4, ::4]), x3_s=t2n(x3[0, :, ::2, ::2]),
         feat_s=t2n(feat[0, :, ::8, ::8]), feat_win=t2n(feat[0, :, 120:128, 152:168]),
         feat_sum=t2n(feat.double().sum((2, 3))[0]),
         heat_s=t2n(heads[0][0, :, ::4, ::4]), size_s=t2n(heads[1][0, :, ::4, ::4]),
         ori_s=t2n(heads[2][0, :, ::4, ::4]), seg_s=t2n(heads[3][0, :, ::4, ::4]),
         head_sum=np.stack([t2n(h.double().sum()) for h in heads]))
    return feat


def gold_planner(lm, up, feat):
    im = ref_mi.InferModel(lm, up, 1.5, 2.4, device=torch.device("cpu"))
    # fixed "others": (X, Y, h, w, cos, sin) in 320x320 pixel space
    det = [(140, 200, 2.0, 4.0, 0.9, 0.1), (190, 150, 2.0, 4.0, -0.3, 0.8), (100, 260, 1.5, 3.0, 0.0, -1.0),
           (160, 279, 2, 4, 1.0, 0.0),  # within 4 px of the ego centre -> skipped (model_inference.py:133)
           (222, 90, 2.0, 4.5, 0.5, 0.5)]
    nxp = torch.tensor([1.5, -12.0])
    outs = {}
    for cmd in (0, 3, 5):
        e, p, c, oc, om = im.uniplanner_infer(feat[0], det, cmd, nxp)
        outs[f"ego_plan_{cmd}"] = t2n(p); outs[f"ego_cast_{cmd}"] = t2n(c)
    outs.update(ego_embd=t2n(e), other_cast=t2n(oc), other_cmds=t2n(om))
    # stages
    crop = ref_mi.crop_feature(feat, torch.zeros(1, 2), torch.zeros(1), 2.0, 96, 0.0, 0.75)
    locs = torch.tensor([[-5.0, -20.0], [7.5, -32.5]]); oris = torch.tensor([0.3, -1.2])
    crop2 = ref_mi.crop_feature(feat.expand(2, -1, -1, -1), locs, oris, 2.0, 96, 0.0, 0.75)
    embd2 = up.lidar_conv_emb(crop2)
    cast_all = up.cast(embd2)
    plan_all = up.plan(embd2, torch.tensor([[1.5, -12.0], [-3.0, -8.0]]), cast_locs=cast_all,
                       pixels_per_meter=4, crop_size=192)
    outs.update(crop_ego_s=t2n(crop[0, ::4, ::3, ::3]), crop2_s=t2n(crop2[:, ::4, ::3, ::3]), embd2=t2n(embd2),
                cast_all=t2n(cast_all), plan_all=t2n(plan_all), cmd_pred2=t2n(up.cast_cmd_pred(embd2)),
                det=np.array(det, np.float64), nxp=t2n(nxp))
    # stand-alone GRU golden from a synthetic embedding (no dependence on the conv stack)
    r = np.random.Generator(np.random.PCG64(5))
    embd = torch.from_numpy(np.abs(r.normal(0.2, 0.3, (3, 512))).astype(np.float32))
    nx = torch.from_numpy(r.uniform(-15, 15, (3, 2)).astype(np.float32))
    c = up.cast(embd)
    outs.update(gru_embd=t2n(embd), gru_nxp=t2n(nx), gru_cast=t2n(c),
                gru_plan=t2n(up.plan(embd, nx, cast_locs=c, pixels_per_meter=4, crop_size=192)),
                gru_cmd=t2n(up.cast_cmd_pred(embd)))
    save("planner", **outs)


def gold_e2e(lm, up):
    im = ref_mi.InferModel(lm, up, 1.5, 2.4, device=torch.device("cpu"))
    out = {}
    # "c" is the agent's own cloud size: 3 sweeps x 65 536 points = 196 608 (lav_agent_fast.py:240-245, 363-383)
    for name, n, kind in (("a", 32768, "lidar"), ("b", 16384, "uniform"), ("c", 65536, "lidar")):
        pts = synth.stacked_lidar(n, kind=kind)
        nxp = {"a": torch.tensor([0.0, -10.0]), "b": torch.tensor([2.5, -14.0]), "c": torch.tensor([-1.5, -12.0])}[name]
        cmd = {"a": 3, "b": 1, "c": 0}[name]
        e, p, c, oc, om, bev, det = im(torch.from_numpy(pts), nxp, cmd)
        out.update({f"{name}/in_crc": np.array([crc(pts)], np.int64), f"{name}/nxp": t2n(nxp),
                    f"{name}/cmd": np.array([cmd]), f"{name}/ego_embd": t2n(e), f"{name}/ego_plan": t2n(p),
                    f"{name}/ego_cast": t2n(c), f"{name}/other_cast": t2n(oc), f"{name}/other_cmds": t2n(om),
                    f"{name}/bev_s": t2n(bev[0, :, ::4, ::4]),
                    f"{name}/det0": np.array(det[0], np.float64).reshape(-1, 6),
                    f"{name}/det1": np.array(det[1], np.float64).reshape(-1, 6)})
    save("e2e", **out)


def gold_rgb():
    seg = RGBSegmentationModel([4, 6, 7, 10]).eval()
    seg.load_state_dict(synth.seeded_state_dict(seg, prefix="seg."))
    bra = RGBBrakePredictionModel([4, 6, 7, 10]).eval()
    bra.load_state_dict(synth.seeded_state_dict(bra, prefix="bra."))
    cams, tel = synth.rgb_frames()
    rgbs = [c[..., :3][..., ::-1] for c in cams]                    # lav_agent_fast.py:252-254
    all_rgb = torch.tensor(np.stack(rgbs, 0).copy()).permute(0, 3, 1, 2).float()
    logits = seg(all_rgb)
    sem = torch.softmax(logits, dim=1)
    rgb = torch.tensor(np.concatenate(rgbs, axis=1)[None].copy()).permute(0, 3, 1, 2).float()
    tel_rgb = tel[..., :3][..., ::-1][:-96].copy()
    tel_t = torch.tensor(tel_rgb[None]).permute(0, 3, 1, 2).float()
    pred_bra = bra(rgb, tel_t)
    # the brake net stage by stage (team_code_v2/models/rgb.py:66-74, lav/models/attention.py:21-38)
    x1 = bra.conv_backbone(bra.normalize(rgb / 255.))
    x2 = bra.conv_backbone(bra.normalize(tel_t / 255.))
    h1, h2 = bra.attn1(x1), bra.attn2(x2)
    logit = bra.classifier[0](torch.cat([h1, h2], dim=1))
    assert torch.equal(torch.sigmoid(logit)[:, 0], pred_bra)
    save("rgb", logits_s=t2n(logits[:, :, ::4, ::4]), sem_s=t2n(sem[:, :, ::4, ::4]),
         logits_sum=t2n(logits.double().sum((2, 3))), pred_bra=t2n(pred_bra), bra_logit=t2n(logit),
         bra_x1_s=t2n(x1[:, ::8]), bra_x2_s=t2n(x2[:, ::8]), bra_h1=t2n(h1), bra_h2=t2n(h2))


def gold_agent():
    """Host glue of the agent: the reference's own ekf.py / pid.py / planner.py / waypointer.py on the seeded scenario."""
    from ekf import EKF  # noqa: E402  (reference)
    from pid import PIDController  # noqa: E402  (reference)
    from planner import RoutePlanner  # noqa: E402  (reference)
    from waypointer import Waypointer  # noqa: E402  (reference)
    from agents.navigation.local_planner import RoadOption
    sc = synth.agent_scenario()
    ekf = EKF(1, 1.477531, 1.393600)
    ekf.init(*sc["ekf_gps"][0], sc["ekf_compass"][0] - np.pi / 2)
    xs = []
    for (spd, steer), (la, lo), comp in zip(sc["ekf_in"], sc["ekf_gps"], sc["ekf_compass"]):
        ekf.step(spd, steer, la, lo, comp - np.pi / 2)
        xs.append(ekf.x.copy())
    pid_a = PIDController(K_P=0.8, K_I=0.5, K_D=0.2, n=40)
    pid_b = PIDController(K_P=5.0, K_I=0.5, K_D=1.0, n=3)
    pa = [pid_a.step(e) for e in sc["pid_err"]]
    pb = [pid_b.step(e) for e in sc["pid_err"]]
    plan = [({"lat": la, "lon": lo, "z": 0.0}, RoadOption(int(c))) for la, lo, c in zip(sc["lat"], sc["lon"], sc["cmds"])]
    rp = RoutePlanner(plan)
    wp = Waypointer(plan, sc["gps"][0], pop_lane_change=True)
    wp2 = Waypointer(plan, sc["gps"][0], pop_lane_change=False, pop_turning=True)
    r_out, w_out, w2_out = [], [], []
    for g in sc["gps"]:
        r_out.append(rp.run_step(g))
        dx, dy, c = wp.tick(g); w_out.append([dx, dy, c.value])
        dx, dy, c = wp2.tick(g); w2_out.append([dx, dy, c.value])
    save("agent", ekf_x=np.array(xs), pid_a=np.array(pa), pid_b=np.array(pb), route=np.array(r_out),
         waypointer=np.array(w_out), waypointer_turn=np.array(w2_out), gps_crc=np.array([crc(sc["gps"])]))


def gold_agent_fast(lm, up, ticks=24, n_points=8192):
    """The REFERENCE AGENT ITSELF (team_code_v2/lav_agent_fast.py) driven over `ticks` leaderboard ticks of
    synth.agent_scenario() on CPU: run_step's controls, the ego-box filter (`preprocess`, :450-457), the temporal
    stacking (`get_stacked_lidar` / `move_lidar_points`, :363-383, 547-565) and what InferModel receives.
    Stand-ins: carla / leaderboard / wandb / cv2 (tests/golden/_shims), the two torch.jit traces and the four
    checkpoints (seeded state_dicts of the reference's own modules), cuda -> cpu, visualize() (debug video only)."""
    import types
    import lav_agent_fast as ref_agent  # noqa: E402  (reference)
    from agents.navigation.local_planner import RoadOption

    seg = RGBSegmentationModel([4, 6, 7, 10]).eval()
    seg.load_state_dict(synth.seeded_state_dict(seg, prefix="seg."))
    bra = RGBBrakePredictionModel([4, 6, 7, 10]).eval()
    bra.load_state_dict(synth.seeded_state_dict(bra, prefix="bra."))
    cpu = torch.device("cpu")

    class TorchProxy:
        """`torch` as lav_agent_fast sees it: device('cuda') -> cpu, load / jit.load -> seeded weights."""
        jit = types.SimpleNamespace(load=lambda path, *a, **k: bra if "bra" in str(path) else seg)
        cuda = types.SimpleNamespace(empty_cache=lambda: None)

        def __getattr__(self, name):
            return getattr(torch, name)

        @staticmethod
        def device(*a, **k):
            return cpu

        @staticmethod
        def load(path, *a, **k):
            m = lm if "lidar" in str(path) else up
            return {k2: v.clone() for k2, v in m.state_dict().items()}

    ref_agent.torch = TorchProxy()
    ref_agent.InferModel = lambda l, u, cx, cz: ref_mi.InferModel(l, u, cx, cz, device=cpu)
    rec = {}

    class Agent(ref_agent.LAVAgent):
        def visualize(self, *a, **k):                      # debug video (OpenCV): not part of the path
            return np.zeros((1, 1, 3), np.uint8)

        def preprocess(self, lidar_xyzr, lidar_painted=None):
            out = super().preprocess(lidar_xyzr, lidar_painted)
            rec["pre_in"], rec["pre_out"] = t2n(lidar_xyzr), t2n(out)
            return out

        def get_stacked_lidar(self):
            out = super().get_stacked_lidar()
            rec["stacked"] = t2n(out)
            rec["pose"] = (np.array(self.locs[-1], np.float64), float(self.oris[-1]))
            return out

    agent = Agent(os.path.join(REF, "team_code_v2", "config.yaml"))
    sc = synth.agent_scenario()
    agent.set_global_plan([({"lat": la, "lon": lo, "z": 0.0}, RoadOption(int(c)))
                           for la, lo, c in zip(sc["lat"], sc["lon"], sc["cmds"])])
    real_infer = agent.infer_model.forward

    def infer(lidar_points, nxps, cmd_value):
        res = real_infer(lidar_points, nxps, cmd_value)
        rec["nxp"], rec["cmd"] = t2n(nxps), int(cmd_value)
        rec["ego_plan"], rec["ego_cast"] = t2n(res[1]), t2n(res[2])
        rec["other_cast"], rec["other_cmds"] = t2n(res[3]), t2n(res[4])
        rec["det1"] = np.array(res[6][1], np.float64).reshape(-1, 6)
        return res
    agent.infer_model.forward = infer
    real_bra = agent.bra_model

    def bra_hook(rgbs, tel):
        out = real_bra(rgbs, tel)
        rec["pred_bra"] = float(out)
        return out
    agent.bra_model = bra_hook

    out = {"ticks": np.array([ticks]), "n_points": np.array([n_points])}
    controls, poses, nxps, cmds, pred_bras, stack_rows, kept_rows = [], [], [], [], [], [], []
    detail = {12, ticks - 1}        # ticks whose full tensors are stored (the stack holds 3 sweeps from tick 11 on)
    for i in range(ticks):
        rec.clear()
        data = synth.agent_inputs(i, sc, n_points=n_points)
        c = agent.run_step(data, i * 0.05)
        controls.append([c.steer, c.throttle, c.brake])
        if "stacked" not in rec:          # first tick: no inference (lav_agent_fast.py:230-232)
            poses.append([np.nan] * 3); nxps.append([np.nan] * 2); cmds.append(-1); pred_bras.append(np.nan)
            stack_rows.append(0); kept_rows.append(0)
            continue
        poses.append([rec["pose"][0][0], rec["pose"][0][1], rec["pose"][1]])
        nxps.append(rec["nxp"]); cmds.append(rec["cmd"]); pred_bras.append(rec["pred_bra"])
        stack_rows.append(len(rec["stacked"])); kept_rows.append(len(rec["pre_out"]))
        out[f"t{i}/stacked_sum"] = rec["stacked"].astype(np.float64).sum(0)      # every tick: a cheap fingerprint
        out[f"t{i}/ego_plan"] = rec["ego_plan"]
        out[f"t{i}/ego_cast"] = rec["ego_cast"]
        out[f"t{i}/det1"] = rec["det1"]
        if i in detail:
            out[f"t{i}/pre_in_crc"] = np.array([crc(rec["pre_in"])], np.int64)
            out[f"t{i}/pre_keep"] = np.packbits(np.isin(np.arange(len(rec["pre_in"])), _kept_index(rec["pre_in"], rec["pre_out"])))
            out[f"t{i}/stacked"] = rec["stacked"].astype(np.float32)
            out[f"t{i}/other_cast"] = rec["other_cast"]
            out[f"t{i}/other_cmds"] = rec["other_cmds"]
        print(f"agent tick {i}: controls {controls[-1]}, stacked {stack_rows[-1]} rows, cmd {cmds[-1]}", flush=True)
    out.update(controls=np.array(controls, np.float64), poses=np.array(poses, np.float64), nxps=np.array(nxps, np.float64),
               cmds=np.array(cmds), pred_bra=np.array(pred_bras, np.float64), stack_rows=np.array(stack_rows),
               kept_rows=np.array(kept_rows))
    # move_lidar_points on its own (lav_agent_fast.py:547-565)
    xyz = torch.from_numpy(synth.lidar_sweep(2000, name="mlp")[:, :3].copy())
    out["mlp_out"] = t2n(ref_agent.move_lidar_points(xyz, np.array([1.25, -0.4]) - np.array([0.5, 0.3]), 0.31, 0.27))
    save("agent_fast", **out)


def _kept_index(full, kept):
    """Row indices of `full` that `kept` (an order-preserving sub-sequence of it) consists of."""
    idx, j = [], 0
    for i in range(len(full)):
        if j < len(kept) and np.array_equal(full[i], kept[j]):
            idx.append(i); j += 1
    assert j == len(kept)
    return np.array(idx)


def gold_train():
    """Two optimisation steps of the REFERENCE trainers (lav/lav_privileged_v2.py `train_bev`, lav/lav_final_v2.py
    `train_lidar`) on CPU, on lav_amd.train.synthetic batches with seeded weights: their loss terms are the fixture
    tests/test_gpu_train.py holds the MI355X training step to.  torch.load is patched to hand the constructors seeded
    state_dicts (the released checkpoints are LFS pointers)."""
    import types
    sys.path.insert(0, REF)
    import lav.lav_privileged_v2 as ref_priv  # noqa: E402  (reference)
    import lav.lav_final_v2 as ref_final  # noqa: E402  (reference)
    import lav_amd
    from lav_amd.train import TrainConfig, synthetic_bev_batch, synthetic_lidar_batch
    from lav_amd.train.lav import LAV as OurLAV
    torch.set_grad_enabled(True)
    ours = OurLAV(TrainConfig(), "cpu", what="bev")          # only to obtain the seeded state_dicts by key
    bev_sd = {k: v.clone() for k, v in ours.bev_planner.state_dict().items()}
    lm = lav_amd.LiDARModel(num_input=16, backbone="cnn", num_features=[64, 64], **CFG)
    lidar_sd = synth.seeded_state_dict(lm, prefix="lidar.")
    y_off = 1 + CFG["min_x"] / ((CFG["max_x"] - CFG["min_x"]) / 2)
    up = lav_amd.UniPlanner(ours.bev_planner, pixels_per_meter=4, crop_size=96, feature_x_jitter=1.5, feature_angle_jitter=20,
                            x_offset=0, y_offset=y_off, num_cmds=6, num_plan=20, num_input_feature=384, num_plan_iter=5)
    uni_sd = synth.seeded_state_dict(up, prefix="uni.")
    uni_sd.update({"bev_planner." + k: v for k, v in bev_sd.items()})
    real_load = torch.load

    def fake_load(path, *a, **k):
        path = str(path)
        return {k2: v.clone() for k2, v in (bev_sd if "bev" in path else lidar_sd if "lidar" in path else uni_sd).items()}

    args = types.SimpleNamespace(config_path=os.path.join(REF, "config_v2.yaml"), device="cpu", lr=3e-4, perceive_only=False,
                                 motion_only=False)
    torch.load = fake_load
    try:
        out = {}
        # the privileged trainer also builds the camera nets (ImageNet download): not part of train_bev, stubbed out
        ref_priv.RGBBrakePredictionModel = lambda *a, **k: torch.nn.Linear(1, 1)
        ref_priv.RGBSegmentationModel = lambda *a, **k: torch.nn.Linear(1, 1)
        trainer = ref_priv.LAV(args)
        trainer.bev_planner.load_state_dict(bev_sd)
        batch = synthetic_bev_batch(2, seed=11, num_objs=3)
        keys = ("plan_loss", "ego_cast_loss", "other_cast_loss", "cmd_loss")
        rows = []
        for step in range(2):
            torch.manual_seed(100 + step)
            info = trainer.train_bev(*batch, other_weight=0.5)
            rows.append([info[k] for k in keys])
        out["bev_terms"] = np.array(rows)
        trainer = ref_final.LAV(args)
        trainer.distill = True     # read by train_lidar but absent from config_v2.yaml; team_code_v2/config.yaml:11 says True
        batch = synthetic_lidar_batch(2, seed=12, max_points=20000, num_objs=3)
        keys = ("hm_loss", "box_loss", "ori_loss", "seg_loss", "plan_loss", "ego_cast_loss", "other_cast_loss", "cmd_loss")
        rows = []
        for step in range(2):
            torch.manual_seed(200 + step)
            info = trainer.train_lidar(*batch)
            rows.append([info[k] for k in keys])
            print("train_lidar step", step, rows[-1], flush=True)
        out["lidar_terms"] = np.array(rows)
        save("train", **out)
    finally:
        torch.load = real_load
        torch.set_grad_enabled(False)


def gold_train_curve(steps=None, name="train_curve", batch=2, points=20000, nbatches=4, seed0=40):
    """BASELINE.json config #5 asks for a loss-curve match over 500 steps: the REFERENCE trainer (lav/lav_final_v2.py
    `train_lidar`, the loop of lav/train_full_v2.py:24-46) on CPU over `steps` optimisation steps, cycling through four
    seeded synthetic batches of 2 samples (20 000-point clouds), global torch seed re-set per step like gold_train.
    Stores every loss term of every step (tests/golden/train_curve.npz).
    `train_curve_b8` (round 4): the same at batch 8 with config #5's own cloud size (120 000 points), two alternating batches,
    100 steps - the largest batch the reference's CPU run finishes in the build container's time budget."""
    import types
    sys.path.insert(0, REF)
    import lav.lav_final_v2 as ref_final  # noqa: E402  (reference)
    import lav_amd
    from lav_amd.train import TrainConfig, synthetic_lidar_batch
    from lav_amd.train.lav import LAV as OurLAV
    steps = steps or int(os.environ.get("LAV_CURVE_STEPS", "500"))
    torch.set_grad_enabled(True)
    ours = OurLAV(TrainConfig(), "cpu", what="bev")
    bev_sd = {k: v.clone() for k, v in ours.bev_planner.state_dict().items()}
    lm = lav_amd.LiDARModel(num_input=16, backbone="cnn", num_features=[64, 64], **CFG)
    lidar_sd = synth.seeded_state_dict(lm, prefix="lidar.")
    y_off = 1 + CFG["min_x"] / ((CFG["max_x"] - CFG["min_x"]) / 2)
    up = lav_amd.UniPlanner(ours.bev_planner, pixels_per_meter=4, crop_size=96, feature_x_jitter=1.5, feature_angle_jitter=20,
                            x_offset=0, y_offset=y_off, num_cmds=6, num_plan=20, num_input_feature=384, num_plan_iter=5)
    uni_sd = synth.seeded_state_dict(up, prefix="uni.")
    uni_sd.update({"bev_planner." + k: v for k, v in bev_sd.items()})
    real_load = torch.load

    def fake_load(path, *a, **k):
        path = str(path)
        return {k2: v.clone() for k2, v in (bev_sd if "bev" in path else lidar_sd if "lidar" in path else uni_sd).items()}

    args = types.SimpleNamespace(config_path=os.path.join(REF, "config_v2.yaml"), device="cpu", lr=3e-4, perceive_only=False,
                                 motion_only=False)
    torch.load = fake_load
    try:
        trainer = ref_final.LAV(args)
        trainer.distill = True
        batches = [synthetic_lidar_batch(batch, seed=seed0 + i, max_points=points, num_objs=3) for i in range(nbatches)]
        keys = ("hm_loss", "box_loss", "ori_loss", "seg_loss", "plan_loss", "ego_cast_loss", "other_cast_loss", "cmd_loss")
        rows = []
        import time
        t0 = time.time()
        meta = dict(keys=np.array(keys), batch=np.array([batch]), points=np.array([points]), nbatches=np.array([nbatches]), seed0=np.array([seed0]))
        for step in range(steps):
            torch.manual_seed(1000 + step)
            info = trainer.train_lidar(*batches[step % nbatches])
            rows.append([info[k] for k in keys])
            if step % 10 == 0:
                print(f"curve step {step}: {[round(v, 4) for v in rows[-1]]}  ({time.time() - t0:.0f} s)", flush=True)
                save(name, terms=np.array(rows), **meta)
        save(name, terms=np.array(rows), **meta)
    finally:
        torch.load = real_load
        torch.set_grad_enabled(False)


from tests.util import DATASET_CASES, dataset_fixture_config  # noqa: E402  (shared with tests/test_data_host.py)


def gold_datasets():
    """Samples of the REFERENCE's data loaders (lav/utils/datasets/*.py, the classes train_bev_v2.py / train_full_v2.py read
    through get_data_loader) on the synthetic routes, with torch / numpy seeded per sample.  The classes run unmodified over
    tests/golden/_shims/{lmdb,cv2,numba}.py; `lmdb` and `cv2` delegate to this repository's restatements
    (lav_amd.data.lmdb_ro / image), so the fixture pins the loaders' logic - keys, actor filtering, frames, augmentation,
    target maps, order of random draws - and not liblmdb's file format or OpenCV's interpolation (parity unpinned there)."""
    import tempfile
    import types
    sys.path.insert(0, REF)
    import lav.utils  # noqa: F401
    pkg = types.ModuleType("lav.utils.datasets")          # skip the package __init__: it imports the RGB loaders' `imgaug`
    pkg.__path__ = [os.path.join(REF, "lav", "utils", "datasets")]
    sys.modules["lav.utils.datasets"] = pkg
    from lav.utils.datasets.bev_dataset import BEVDataset
    from lav.utils.datasets.lidar_dataset import LiDARDataset
    from lav.utils.datasets.lidar_painted_dataset import LiDARPaintedDataset
    from lav.utils.datasets.temporal_bev_dataset import TemporalBEVDataset
    from lav.utils.datasets.temporal_lidar_painted_dataset import TemporalLiDARPaintedDataset
    classes = dict(bev=BEVDataset, temporal_bev=TemporalBEVDataset, lidar=LiDARDataset, lidar_painted=LiDARPaintedDataset,
                   temporal_lidar_painted=TemporalLiDARPaintedDataset)
    out = {}
    with tempfile.TemporaryDirectory() as root:
        cfg_path = dataset_fixture_config(root)
        for name, picks in DATASET_CASES:
            ds = classes[name](cfg_path)
            out[f"{name}/len"] = len(ds)
            where = {(os.path.basename(ds.dir_map[i]), ds.idx_map[i]): i for i in range(len(ds))}    # the reference walks routes in glob order
            for route, frame in [("route_000", p) if p < 10 else ("route_001", p - 10) for p in picks]:
                i = where[(route, frame)]
                torch.manual_seed(1000 + frame)
                np.random.seed(1000 + frame)
                with torch.enable_grad():
                    sample = ds[i]
                for k, v in enumerate(sample):
                    v = np.asarray(v)
                    if name in ("lidar", "lidar_painted") and k == 0:
                        v = v[:sample[1]]                       # the tail of these two loaders' point buffer is np.empty
                    out[f"{name}/{route}/{frame}/{k}"] = v
    save("datasets", **out)


def gold_keys(lm, up):
    import json
    seg = RGBSegmentationModel([4, 6, 7, 10]); bra = RGBBrakePredictionModel([4, 6, 7, 10])
    d = {n: [[k, list(v.shape)] for k, v in m.state_dict().items()] for n, m in
         dict(lidar=lm, uniplanner=up, seg=seg, bra=bra).items()}
    json.dump(d, open(os.path.join(HERE, "state_dict_keys.json"), "w"))
    print("state_dict_keys.json", sum(len(v) for v in d.values()), "keys")


if __name__ == "__main__":
    if sys.argv[1:] == ["agent"]:     # only the host-glue fixture
        gold_agent()
        sys.exit(0)
    if sys.argv[1:] == ["e2e"]:     # only the end-to-end fixture
        lm, up = build_reference()
        gold_e2e(lm, up)
        sys.exit(0)
    if sys.argv[1:] == ["rgb"]:     # only the camera-net fixture
        gold_rgb()
        sys.exit(0)
    if sys.argv[1:] == ["agent_fast"]:     # only the reference-agent fixture
        lm, up = build_reference()
        gold_agent_fast(lm, up)
        sys.exit(0)
    if sys.argv[1:] == ["train_curve"]:     # only the loss-curve fixture (hours of CPU time)
        gold_train_curve()
        sys.exit(0)
    if sys.argv[1:] == ["train_curve_b8"]:     # config #5's cloud size at batch 8, 100 steps (about an hour of CPU time)
        gold_train_curve(steps=int(os.environ.get("LAV_CURVE_STEPS", "100")), name="train_curve_b8", batch=8, points=120000, nbatches=2, seed0=60)
        sys.exit(0)
    if sys.argv[1:] == ["train_curve_b32"]:    # config #5 AS STATED: batch 32 x 120 000-point clouds, 25 steps (round 5; ~4 min of CPU per step)
        gold_train_curve(steps=int(os.environ.get("LAV_CURVE_STEPS", "25")), name="train_curve_b32", batch=32, points=120000, nbatches=1, seed0=80)
        sys.exit(0)
    if sys.argv[1:] == ["datasets"]:     # only the data-loader fixture
        gold_datasets()
        sys.exit(0)
    if sys.argv[1:] == ["train"]:     # only the training fixture
        gold_train()
        sys.exit(0)
    lm, up = build_reference()
    gold_keys(lm, up)
    gold_pillar(lm)
    gold_paint(lm, up)
    feat = gold_bev(lm, up)
    gold_planner(lm, up, feat)
    gold_e2e(lm, up)
    gold_rgb()
    gold_agent()
    gold_agent_fast(lm, up)
