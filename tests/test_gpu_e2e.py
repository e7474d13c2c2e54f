"""End-to-end parity on the GPU: BEV stack vs the reference goldens, InferModel.forward waypoints within 1e-4
(BASELINE.json configs #1/#3), and the frame pipeline's first steps."""
import numpy as np
import pytest
import torch

import lav_amd
from lav_amd import synth
from tests.util import assert_close, build_models, crc

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda")


@pytest.fixture(scope="module")
def models():
    return build_models(DEV)


def test_backbone_heads_vs_reference_golden(golden, models):
    g = golden["bev"]
    lm, _ = models
    pts = synth.stacked_lidar(8192)
    assert crc(pts) == int(g["in_crc"][0])
    with torch.no_grad():
        canvas = lm.point_pillar_net([torch.from_numpy(pts).to(DEV)], [len(pts)])
        assert_close(canvas.double().sum((2, 3))[0].cpu().numpy(), g["canvas_sum"], atol=1e-2, rtol=1e-5, what="canvas sums")
        feat = lm.backbone(canvas)
        heads = lm.heads(feat)
        single = [h(feat) for h in (lm.center_head, lm.box_head, lm.ori_head, lm.seg_head)]
    f = feat.cpu()
    assert_close(f[0, :, ::8, ::8].numpy(), g["feat_s"], atol=3e-5, rtol=1e-4, what="features (strided)")
    assert_close(f[0, :, 120:128, 152:168].numpy(), g["feat_win"], atol=3e-5, rtol=1e-4, what="features (window)")
    assert_close(f.double().sum((2, 3))[0].numpy(), g["feat_sum"], atol=5e-2, rtol=1e-4, what="feature sums")
    for h, s, key in zip(heads, single, ("heat_s", "size_s", "ori_s", "seg_s")):
        assert_close(h[0, :, ::4, ::4].cpu().numpy(), g[key], atol=3e-5, rtol=1e-4, what=key)
        assert_close(s.cpu().numpy(), h.cpu().numpy(), atol=1e-5, what="fused heads == per-head modules")


def test_planner_stages_vs_reference_golden(golden, models):
    g = golden["planner"]
    lm, up = models
    pts = synth.stacked_lidar(8192)
    im = lav_amd.InferModel(lm, up, 1.5, 2.4, device=DEV)
    with torch.no_grad():
        feat = lm.backbone(lm.point_pillar_net([torch.from_numpy(pts).to(DEV)], [len(pts)]))
        det = [tuple(r) for r in g["det"]]
        nxp = torch.from_numpy(g["nxp"]).to(DEV)
        for cmd in (0, 3, 5):
            e, p, c, oc, om = im.uniplanner_infer(feat[0], det, cmd, nxp)
            assert_close(p.cpu().numpy(), g[f"ego_plan_{cmd}"], atol=1e-4, what=f"ego plan cmd {cmd}")
            assert_close(c.cpu().numpy(), g[f"ego_cast_{cmd}"], atol=1e-4, what=f"ego cast cmd {cmd}")
        assert_close(e.cpu().numpy(), g["ego_embd"], atol=2e-5, rtol=1e-4, what="ego embedding")
        assert_close(oc.cpu().numpy(), g["other_cast"], atol=1e-4, what="others' forecasts")
        assert_close(om.cpu().numpy(), g["other_cmds"], atol=1e-5, what="others' command scores")
        locs = torch.tensor([[-5.0, -20.0], [7.5, -32.5]], device=DEV)
        oris = torch.tensor([0.3, -1.2], device=DEV)
        crop2 = up.crop_feature(feat.expand(2, -1, -1, -1), locs, oris, pixels_per_meter=2.0, crop_size=96)
        assert_close(crop2[:, ::4, ::3, ::3].cpu().numpy(), g["crop2_s"], atol=3e-5, rtol=1e-4, what="rotated crops")
        embd2 = up.lidar_conv_emb(crop2)
        assert_close(embd2.cpu().numpy(), g["embd2"], atol=3e-5, rtol=1e-4, what="ResNet-18 embeddings")


@pytest.mark.parametrize("name,n,kind", [("a", 32768, "lidar"), ("b", 16384, "uniform"), ("c", 65536, "lidar")])
def test_infer_model_forward_vs_reference_golden(golden, models, name, n, kind):
    """Full InferModel.forward: waypoints within 1e-4 of the reference PyTorch path (BASELINE.json)."""
    g = golden["e2e"]
    lm, up = models
    im = lav_amd.InferModel(lm, up, 1.5, 2.4, device=DEV)
    pts = synth.stacked_lidar(n, kind=kind)
    assert crc(pts) == int(g[f"{name}/in_crc"][0])
    e, p, c, oc, om, bev, det = im(torch.from_numpy(pts).to(DEV), torch.from_numpy(g[f"{name}/nxp"]).to(DEV), int(g[f"{name}/cmd"][0]))
    for i in (0, 1):
        ref = g[f"{name}/det{i}"]
        assert len(det[i]) == len(ref), f"class {i}: {len(det[i])} detections vs {len(ref)}"
        if len(ref):
            np.testing.assert_array_equal(np.array(det[i])[:, :2], ref[:, :2])
            assert_close(np.array(det[i])[:, 2:], ref[:, 2:], atol=1e-4, what="detection attributes")
    assert_close(p.cpu().numpy(), g[f"{name}/ego_plan"], atol=1e-4, what="ego_plan_locs")
    assert_close(c.cpu().numpy(), g[f"{name}/ego_cast"], atol=1e-4, what="ego_cast_locs")
    assert_close(oc.cpu().numpy(), g[f"{name}/other_cast"], atol=1e-4, what="other_cast_locs")
    assert_close(om.cpu().numpy(), g[f"{name}/other_cmds"], atol=1e-5, what="other_cast_cmds")
    assert_close(e.cpu().numpy(), g[f"{name}/ego_embd"], atol=3e-5, rtol=1e-4, what="ego_embd")
    assert_close(bev[0, :, ::4, ::4].cpu().numpy(), g[f"{name}/bev_s"], atol=1e-5, what="pred_bev")
    if len(g[f"{name}/other_cast"]) == 0:
        assert oc.device.type == "cpu" and om.device.type == "cpu"   # reference returns CPU zeros (model_inference.py:167-168)


def test_camera_nets_vs_reference_golden(golden):
    """ERFNet (logits and softmax) and the brake net stage by stage - both ResNet-18 trunks, both attention poolings,
    the logit and the sigmoid - against the REFERENCE's modules (tests/golden/rgb.npz; team_code_v2/models/rgb.py:36-83,
    lav/models/attention.py:21-38)."""
    from lav_amd.rgb import RGBBrakePredictionModel, RGBSegmentationModel
    g = golden["rgb"]
    seg = RGBSegmentationModel([4, 6, 7, 10]); seg.load_state_dict(synth.seeded_state_dict(seg, prefix="seg.")); seg.eval()
    bra = RGBBrakePredictionModel([4, 6, 7, 10]); bra.load_state_dict(synth.seeded_state_dict(bra, prefix="bra.")); bra.eval()
    cams, tel = synth.rgb_frames()
    rgbs = [c[..., :3][..., ::-1] for c in cams]
    all_rgb = torch.tensor(np.stack(rgbs, 0).copy()).permute(0, 3, 1, 2).float()
    wide = torch.tensor(np.concatenate(rgbs, axis=1)[None].copy()).permute(0, 3, 1, 2).float()
    tel_rgb = torch.tensor(tel[..., :3][..., ::-1][:-96][None].copy()).permute(0, 3, 1, 2).float()
    with torch.no_grad():
        seg.to(DEV); bra.to(DEV)
        logits = seg(all_rgb.to(DEV))
        assert_close(logits[:, :, ::4, ::4].cpu().numpy(), g["logits_s"], atol=1e-5 * float(np.abs(g["logits_s"]).max()), rtol=1e-4, what="ERFNet logits")  # seeded weights give |logit| ~ 2e3
        assert_close(logits.double().sum((2, 3)).cpu().numpy(), g["logits_sum"], atol=0.5, rtol=1e-4, what="ERFNet logit sums")
        sem = torch.softmax(logits, dim=1)
        # softmax is 1/2-Lipschitz in the max norm of the logits: with the seeded weights' |logit| ~ 2e3 the logit bar above
        # (1e-5 relative) allows half that much here
        assert_close(sem[:, :, ::4, ::4].cpu().numpy(), g["sem_s"], atol=0.5e-5 * float(np.abs(g["logits_s"]).max()), what="softmax(ERFNet)")
        x1 = bra.conv_backbone(bra.normalize(wide.to(DEV) / 255.))
        x2 = bra.conv_backbone(bra.normalize(tel_rgb.to(DEV) / 255.))
        scale = float(np.abs(g["bra_x1_s"]).max())
        assert_close(x1[:, ::8].cpu().numpy(), g["bra_x1_s"], atol=1e-5 * scale, rtol=1e-4, what="brake trunk (3 views)")
        assert_close(x2[:, ::8].cpu().numpy(), g["bra_x2_s"], atol=1e-5 * scale, rtol=1e-4, what="brake trunk (tele)")
        h1, h2 = bra.attn1(x1), bra.attn2(x2)
        hs = float(np.abs(g["bra_h1"]).max())
        assert_close(h1.cpu().numpy(), g["bra_h1"], atol=1e-5 * hs, rtol=1e-4, what="attention pooling 1")
        assert_close(h2.cpu().numpy(), g["bra_h2"], atol=1e-5 * hs, rtol=1e-4, what="attention pooling 2")
        logit = bra.classifier[0](torch.cat([h1, h2], dim=1))
        assert_close(logit.cpu().numpy(), g["bra_logit"], atol=1e-4, what="brake logit")
        assert_close(bra(wide.to(DEV), tel_rgb.to(DEV)).cpu().numpy(), g["pred_bra"], atol=1e-5, what="pred_bra")


def test_graphed_frame_pipeline_matches_eager(models):
    """HIP-graph replay of the frame (static buffers, NaN padding, ring history) == the eager FramePipeline
    over 18 frames (history fill, ring wrap-around, varying poses, two command values)."""
    from lav_amd.frame import FramePipeline, GraphedFramePipeline
    from lav_amd.rgb import RGBBrakePredictionModel, RGBSegmentationModel
    lm, up = models
    seg = RGBSegmentationModel([4, 6, 7, 10]); seg.load_state_dict(synth.seeded_state_dict(seg, prefix="seg.")); seg.eval().to(DEV)
    bra = RGBBrakePredictionModel([4, 6, 7, 10]); bra.load_state_dict(synth.seeded_state_dict(bra, prefix="bra.")); bra.eval().to(DEV)
    eager = FramePipeline(lm, up, seg, bra, 1.5, 2.4, device=DEV)
    graph = GraphedFramePipeline(lm, up, seg, bra, 1.5, 2.4, device=DEV, points_per_tick=8192)
    cams, tel = synth.rgb_frames()
    rgbs = [c[..., :3][..., ::-1] for c in cams]
    all_rgb = torch.tensor(np.stack(rgbs, 0).copy()).permute(0, 3, 1, 2).float().to(DEV)
    wide = torch.tensor(np.concatenate(rgbs, axis=1)[None].copy()).permute(0, 3, 1, 2).float().to(DEV)
    tel_rgb = torch.tensor(tel[..., :3][..., ::-1][:-96][None].copy()).permute(0, 3, 1, 2).float().to(DEV)
    nxp = torch.tensor([1.0, -9.0], device=DEV)
    for i in range(19):
        n = 8192 if i % 3 else 7000                       # ticks shorter than the static buffer are NaN padded
        tick = torch.from_numpy(synth.lidar_sweep(n, name=f"g{i}")).to(DEV)
        loc, ori = np.array([0.3 * i, 0.05 * i]), 0.02 * i
        cmd = 3 if i < 12 else 1
        a = eager.step(tick, all_rgb, wide, tel_rgb, loc, ori, nxp, cmd)
        b = graph.step(tick, all_rgb, wide, tel_rgb, loc, ori, nxp, cmd)
        if a is None:
            assert b is None
            continue
        # the eager path stacks sweeps with a torch matmul, the graphed one with lav_stack_sweeps: xyz may differ in the
        # last bit, so pixel positions must agree exactly and the regressed maps within float noise
        for c in range(2):
            assert [d[:2] for d in a["det"][c]] == [d[:2] for d in b["det"][c]], f"frame {i} class {c}"
            assert_close(np.array([d[2:] for d in b["det"][c]]), np.array([d[2:] for d in a["det"][c]]), atol=1e-5,
                         what=f"frame {i} det {c}")
        for k in ("ego_plan_locs", "ego_cast_locs", "other_cast_locs", "other_cast_cmds", "pred_bra"):
            assert_close(b[k].cpu().numpy(), a[k].cpu().numpy(), atol=2e-5, rtol=2e-6, what=f"frame {i} {k}")
        assert_close(b["pred_bev"].cpu().numpy(), a["pred_bev"].cpu().numpy(), atol=1e-5, what=f"frame {i} pred_bev")


def test_host_sensor_tensors_give_the_resident_inputs_frame(models):
    """step() handed HOST tensors (the sensor path of a real drive: pinned channels-last camera tensors, a pageable contiguous LiDAR
    tick) stages them to the device with their strides kept and lets copy_many change the layout there (round 5: copied straight into
    the static buffers, torch permuted the camera tensors on the CPU - 37 ms per frame).  Same frames, same outputs, bit for bit."""
    from lav_amd.frame import GraphedFramePipeline
    from lav_amd.rgb import RGBBrakePredictionModel, RGBSegmentationModel
    lm, up = models
    seg = RGBSegmentationModel([4, 6, 7, 10]); seg.load_state_dict(synth.seeded_state_dict(seg, prefix="seg.")); seg.eval().to(DEV)
    bra = RGBBrakePredictionModel([4, 6, 7, 10]); bra.load_state_dict(synth.seeded_state_dict(bra, prefix="bra.")); bra.eval().to(DEV)
    cams, tel = synth.rgb_frames()
    rgbs = [c[..., :3][..., ::-1] for c in cams]
    # channels-last in memory, as bench.py's synthetic_inputs builds them (transpose + astype keeps the strides)
    h_all = torch.from_numpy(np.stack(rgbs, 0).transpose(0, 3, 1, 2).astype(np.float32)).pin_memory()
    h_wide = torch.from_numpy(np.concatenate(rgbs, axis=1)[None].transpose(0, 3, 1, 2).astype(np.float32)).pin_memory()
    h_tel = torch.from_numpy(tel[..., :3][..., ::-1][:-96][None].transpose(0, 3, 1, 2).astype(np.float32)).pin_memory()
    assert not h_all.is_contiguous() and h_all.is_pinned()
    h_nxp = torch.tensor([1.0, -9.0])
    outs = []
    for host in (False, True):
        pipe = GraphedFramePipeline(lm, up, seg, bra, 1.5, 2.4, device=DEV, points_per_tick=8192)
        o = None
        for i in range(18):
            tick = torch.from_numpy(synth.lidar_sweep(8192 if i % 3 else 7000, name=f"h{i}"))
            args = (tick, h_all, h_wide, h_tel) if host else (tick.to(DEV), h_all.to(DEV), h_wide.to(DEV), h_tel.to(DEV))
            o = pipe.step(*args, np.array([0.3 * i, 0.05 * i]), 0.02 * i, h_nxp if host else h_nxp.to(DEV), 3)
        outs.append({k: o[k].clone() for k in ("ego_plan_locs", "ego_cast_locs", "other_cast_locs", "pred_bra", "pred_bev")})
    for k in outs[0]:
        assert torch.equal(outs[0][k], outs[1][k]), k


def test_forced_others_hook_runs_the_others_branch_on_the_given_poses(models):
    """set_forced_others (frame time at a fixed number of other vehicles): the others branch's outputs are those of the
    given poses - crop_feature + embedder + cast on this frame's feature map - whatever the heads detect; n = 0 gives the
    reference's empty outputs; None restores the detections."""
    from lav_amd.frame import GraphedFramePipeline
    from lav_amd.rgb import RGBBrakePredictionModel, RGBSegmentationModel
    lm, up = models
    seg = RGBSegmentationModel([4, 6, 7, 10]); seg.load_state_dict(synth.seeded_state_dict(seg, prefix="seg.")); seg.eval().to(DEV)
    bra = RGBBrakePredictionModel([4, 6, 7, 10]); bra.load_state_dict(synth.seeded_state_dict(bra, prefix="bra.")); bra.eval().to(DEV)
    pipe = GraphedFramePipeline(lm, up, seg, bra, 1.5, 2.4, device=DEV, points_per_tick=8192)
    cams, tel = synth.rgb_frames()
    rgbs = [c[..., :3][..., ::-1] for c in cams]
    all_rgb = torch.tensor(np.stack(rgbs, 0).copy()).permute(0, 3, 1, 2).float().to(DEV)
    wide = torch.tensor(np.concatenate(rgbs, axis=1)[None].copy()).permute(0, 3, 1, 2).float().to(DEV)
    tel_rgb = torch.tensor(tel[..., :3][..., ::-1][:-96][None].copy()).permute(0, 3, 1, 2).float().to(DEV)
    nxp = torch.tensor([1.0, -9.0], device=DEV)
    step = lambda i: pipe.step(torch.from_numpy(synth.lidar_sweep(8192, name=f"f{i}")).to(DEV), all_rgb, wide, tel_rgb,
                               np.array([0.3 * i, 0.05 * i]), 0.02 * i, nxp, 3)
    locs = np.array([[4.0, -12.0], [-3.5, -20.0], [0.5, 8.0], [7.0, -30.0]], np.float32)
    oris = np.array([0.1, -0.2, 3.0, 0.0], np.float32)
    pipe.set_forced_others(locs, oris)
    for i in range(3):
        out = step(i)
    assert out["other_cast_locs"].shape[:1] == (4,) and out["other_cast_cmds"].shape == (4, up.num_cmds)
    dl, do = torch.from_numpy(locs).to(DEV), torch.from_numpy(oris).to(DEV)
    crops = up.crop_feature(pipe.b_features.expand(4, -1, -1, -1), dl, do, up.pixels_per_meter / 2, up.crop_size)
    _, cast, cmds = up.embed_cast(crops, oris=do, locs=dl, want_cmds=True)
    # (the capacity-15 graph and this 4-row call get different convolution plans: float summation order)
    assert_close(out["other_cast_locs"].cpu().numpy(), cast.cpu().numpy(), atol=2e-5, rtol=2e-6, what="forced others cast")
    assert_close(out["other_cast_cmds"].cpu().numpy(), cmds.cpu().numpy(), atol=2e-6, what="forced others cmds")
    pipe.set_forced_others(locs[:0], oris[:0])
    out0 = step(3)
    assert out0["other_cast_locs"].shape[0] == 0 and out0["other_cast_cmds"].shape[0] == 0
    pipe.set_forced_others(None)
    free = step(4)
    assert free["other_cast_locs"].shape[0] == min(len(up.others_from_detections(free["det"][1], 320, 320)[0]), 15)


def test_erfnet_runs_share_one_cleaning_of_their_counters(monkeypatch):
    """Round 6: ERFNet's four persistent runs of a pass keep their progress counters in regions of one array that the FIRST run zeroes for
    all of them (lav_conv1d_pair_chain_region) instead of a small launch in front of every run: bit-identical to every run cleaning its
    own, pass after pass (each pass cleans again), in both precisions, eagerly and replayed from a HIP graph; no workgroup times out and
    every run is counted."""
    from lav_amd import _lib, ops
    from lav_amd.rgb import RGBSegmentationModel
    seg = RGBSegmentationModel([4, 6, 7, 10]); seg.load_state_dict(synth.seeded_state_dict(seg, prefix="seg.")); seg.eval().to(DEV)
    cams, _ = synth.rgb_frames()
    x = torch.tensor(np.stack([c[..., :3][..., ::-1] for c in cams], 0).copy()).permute(0, 3, 1, 2).float().to(DEV)
    for prec in (_lib.CONV_F16X3, _lib.CONV_BF16X6):
        with torch.no_grad(), ops.precision(prec):
            monkeypatch.setenv("LAV_CHAIN_SHARED_CLEAN", "0")
            want = seg(x).clone()
            monkeypatch.setenv("LAV_CHAIN_SHARED_CLEAN", "1")
            t0, n0 = ops.pair_chain_status(DEV)
            got = [seg(x).clone() for _ in range(3)]
            t1, n1 = ops.pair_chain_status(DEV)
            assert t1 == t0 and n1 == n0 + 3 * 4, (t0, n0, t1, n1)
            assert all(torch.equal(g, want) for g in got)
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                seg(x)                      # (workspace of the capture stream)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                y = seg(x)
            for _ in range(3):
                g.replay()
                torch.cuda.synchronize()
                assert torch.equal(y, want)
            with torch.cuda.stream(s):
                assert ops.pair_chain_status(DEV)[0] == 0
