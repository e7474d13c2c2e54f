"""Parity of the paint and GRU kernels (through the C ABI) against the oracle and the reference goldens."""
import numpy as np
import pytest
import torch

import lav_amd
from lav_amd import ops, synth
from oracle import bev as obev
from oracle import paint as opaint
from tests.util import assert_close, build_models, state_dicts

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda")


@pytest.fixture(scope="module")
def models():
    return build_models(DEV)


def test_paint_bit_exact_vs_oracle_and_reference(golden, models):
    g = golden["paint"]
    lm, up = models
    im = lav_amd.InferModel(lm, up, 1.5, 2.4, device=DEV)
    sem = synth.semantic_probs()
    lidar = g["lidar"]
    fused = im.forward_paint(torch.from_numpy(lidar).to(DEV), torch.from_numpy(sem).to(DEV)).cpu().numpy()
    np.testing.assert_array_equal(fused, opaint.forward_paint(lidar, sem))          # bit-exact vs the oracle
    ref = g["fused"]
    np.testing.assert_array_equal(fused[:, :4], ref[:, :4])
    differ = (fused[:, 4:] != ref[:, 4:]).any(axis=1)
    assert differ.mean() < 2e-3, f"{differ.sum()} painted rows differ from the reference run"
    for i, cc in enumerate(im.coord_converters):
        np.testing.assert_array_equal(cc.K.numpy(), g[f"K{i}"])
        np.testing.assert_array_equal(cc.lidar_to_world.numpy(), g[f"l2w{i}"])
        np.testing.assert_array_equal(cc.world_to_cam.numpy(), g[f"w2c{i}"])
        uvz = cc.to(DEV)(torch.from_numpy(lidar).to(DEV)).cpu().numpy()
        o, raw = opaint.project(lidar[:, :3], *opaint.camera_matrices(opaint.CAMERA_YAWS[i], (0, 0, 2.4), (1.5, 0, 2.4)))
        inside = (np.abs(raw) < 2e9).all(axis=1) & np.isfinite(raw).all(axis=1)
        np.testing.assert_array_equal(uvz[inside], o[inside])


def test_paint_large_and_edge(models):
    lm, up = models
    im = lav_amd.InferModel(lm, up, 1.5, 2.4, device=DEV)
    sem = synth.semantic_probs(seed=3)
    lidar = np.concatenate([synth.lidar_sweep(65536, seed=5), np.array([[np.nan, 0, 0, 1], [np.inf, 1, 1, 1], [1.5, 0, 0, 0]], np.float32)])
    fused = im.forward_paint(torch.from_numpy(lidar).to(DEV), torch.from_numpy(sem).to(DEV)).cpu().numpy()
    o = opaint.forward_paint(lidar, sem)
    np.testing.assert_array_equal(np.nan_to_num(fused, nan=-7.0), np.nan_to_num(o, nan=-7.0))
    empty = im.forward_paint(torch.zeros((0, 4), device=DEV), torch.from_numpy(sem).to(DEV))
    assert empty.shape == (0, 8)


@pytest.mark.parametrize("impl", ["persistent", "lds", "steps"])
def test_gru_cast_plan_vs_reference_golden(golden, models, impl, monkeypatch):
    """plan: one persistent launch with granule all-gathers (default, used when <= 6 state rows: round 5's one-wave-per-workgroup
    kernel without LDS; LAV_PLAN_IMPL=lds: rounds 2-4's four-wave kernel) and the step-per-launch variant (LAV_PLAN_IMPL=steps,
    any batch)."""
    monkeypatch.setenv("LAV_PLAN_IMPL", impl)
    g = golden["planner"]
    _, up = models
    embd = torch.from_numpy(g["gru_embd"]).to(DEV)
    nxp = torch.from_numpy(g["gru_nxp"]).to(DEV)
    cast = up.cast(embd)
    assert_close(cast.cpu().numpy(), g["gru_cast"], atol=2e-5, what="cast")
    plan = up.plan(embd, nxp, cast_locs=cast, pixels_per_meter=4, crop_size=192)
    assert_close(plan.cpu().numpy(), g["gru_plan"], atol=1e-4, what="plan (all commands)")
    for cmd in (0, 3, 5):  # single-branch evaluation == that slice of the full result (3 state rows: persistent kernel)
        one = up.plan(embd, nxp, cast_locs=cast, pixels_per_meter=4, crop_size=192, cmd=cmd)
        assert_close(one[:, :, 0].cpu().numpy(), g["gru_plan"][:, :, cmd], atol=1e-4, what=f"plan cmd {cmd}")
    one = up.plan(embd[:1], nxp[:1], cast_locs=cast[:1], pixels_per_meter=4, crop_size=192, cmd=-1)   # 6 rows
    assert_close(one.cpu().numpy(), g["gru_plan"][:1], atol=1e-4, what="plan B=1 all commands")
    for _ in range(3):  # repeated launches reuse the granule buffers (re-zeroed by the launch)
        again = up.plan(embd[:1], nxp[:1], cast_locs=cast[:1], pixels_per_meter=4, crop_size=192, cmd=3)
        assert_close(again[:, :, 0].cpu().numpy(), g["gru_plan"][:1, :, 3], atol=1e-4, what="plan repeat")
    with torch.no_grad():
        assert_close(up.cast_cmd_pred(embd).cpu().numpy(), g["gru_cmd"], atol=1e-6, what="cmd")


def test_gru_vs_oracle_other_batch(models):
    _, up = models
    _, usd = state_dicts()
    r = np.random.Generator(np.random.PCG64(9))
    embd = torch.from_numpy(np.abs(r.normal(0.2, 0.3, (15, 512))).astype(np.float32))
    with torch.no_grad():
        ref = obev.cast(embd, usd)
    assert_close(up.cast(embd.to(DEV)).cpu().numpy(), ref.numpy(), atol=2e-5, what="cast B=15")
    assert up.cast(torch.zeros((0, 512), device=DEV)).shape == (0, 6, 20, 2)
    # plan with many state rows (15 samples x 6 commands = 90: the trainer's frozen teacher): the MFMA step path, against the
    # oracle and against the VALU step kernel
    nxp = torch.from_numpy(r.normal(0.0, 10.0, (15, 2)).astype(np.float32))
    with torch.no_grad():
        ref_plan = obev.plan(embd, nxp, ref, usd)
    cast = up.cast(embd.to(DEV))
    got = up.plan(embd.to(DEV), nxp.to(DEV), cast_locs=cast, pixels_per_meter=4, crop_size=192)
    assert_close(got.cpu().numpy(), ref_plan.numpy(), atol=1e-4, what="plan B=15, all commands (MFMA steps)")
    one = up.plan(embd.to(DEV), nxp.to(DEV), cast_locs=cast, pixels_per_meter=4, crop_size=192, cmd=2)     # 15 rows: VALU steps
    assert_close(one[:, :, 0].cpu().numpy(), got[:, :, 2].cpu().numpy(), atol=2e-5, what="plan B=15, one command")


def test_plan_timeout_is_loud_and_recoverable(golden, monkeypatch):
    """A persistent plan launch that cannot complete (forced here by a spin limit of 1: a workgroup gives up the first time
    a peer's hidden state has not arrived yet) must never look like a plan: the whole output is NaN, the status word of the
    workspace says so, and the step-per-launch entry point recomputes the reference's waypoints."""
    g = golden["planner"]
    _, up = build_models(DEV)
    embd = torch.from_numpy(g["ego_embd"]).to(DEV).view(1, -1)
    nxp = torch.from_numpy(g["nxp"]).to(DEV).view(1, 2)
    cast = up.cast(embd, mode="ego")
    good = up.plan(embd, nxp, cast_locs=cast, pixels_per_meter=4, crop_size=192, cmd=3)
    torch.cuda.synchronize()
    assert ops.gru_plan_status(1, 512, 6, 3, DEV) == 0 and torch.isfinite(good).all()
    monkeypatch.setenv("LAV_PLAN_SPIN_LIMIT", "1")
    bad = up.plan(embd, nxp, cast_locs=cast, pixels_per_meter=4, crop_size=192, cmd=3)
    torch.cuda.synchronize()
    monkeypatch.delenv("LAV_PLAN_SPIN_LIMIT")
    assert ops.gru_plan_status(1, 512, 6, 3, DEV) == 1, "a launch that gave up must raise the status word"
    assert torch.isnan(bad).all(), "and must not return anything that could pass for waypoints"
    d = ops.gru_plan_diag(1, 512, 6, 3, DEV)   # who gave up, where: all 64 workgroups entered, the first to abort left its trace
    assert d["status"] == 1 and d["entered"] == 64 and 1 <= d["abort_wg_plus1"] <= 64 and d["abort_epoch"] >= 1 and d["abort_spins"] >= 1, d
    assert d["tag_seen"] != d["abort_epoch"] and d["aborted_launches"] >= 1
    again = up.plan(embd, nxp, cast_locs=cast, pixels_per_meter=4, crop_size=192, cmd=3, impl="steps")
    torch.cuda.synchronize()
    assert_close(again.cpu().numpy(), good.cpu().numpy(), atol=2e-5, what="step path after an abort")
    back = up.plan(embd, nxp, cast_locs=cast, pixels_per_meter=4, crop_size=192, cmd=3)
    torch.cuda.synchronize()
    assert ops.gru_plan_status(1, 512, 6, 3, DEV) == 0 and torch.equal(back, good)
    d2 = ops.gru_plan_diag(1, 512, 6, 3, DEV)
    assert d2["status"] == 0 and d2["entered"] == 64 and d2["completed"] == 64 and d2["abort_wg_plus1"] == 0, d2
    assert d2["launches"] >= d["launches"] + 1 and d2["aborted_launches"] == d["aborted_launches"]


@pytest.mark.parametrize("graph", [False, True], ids=["eager", "graph"])
def test_persistent_plan_is_exact_beside_the_crop_stem(graph):
    """The one-launch plan kernel against the step-per-launch path while another stream keeps every CU busy with the others
    branch's 7x7 stem (tap-pair split kernel: 150 KB of LDS per workgroup, its waves share SIMDs with the plan kernel's) -
    the load under which round 4's quarter-poll variant returned finite but wrong plans on every launch (it passed every
    quiet-chip test; tools/plan_stress.py).  Two alternating inputs, so that state left over from the previous launch would
    show; every launch must equal the step path bit for bit and none may abort."""
    from lav_amd import _lib
    dev = torch.device("cuda")
    torch.manual_seed(0)
    H, T, NC, N = 512, 20, 6, 80
    g = lambda *s, sc=1.0: (torch.randn(*s) * sc).to(dev)
    w_ih, w_hh, b_ih, b_hh = g(3 * H, 4, sc=0.3), g(3 * H, H, sc=H ** -0.5), g(3 * H, sc=0.1), g(3 * H, sc=0.1)
    mlp_w, mlp_b = g(2, H, sc=0.05), g(2, sc=0.1)
    inputs = [(g(1, H, sc=0.5), g(1, 2, sc=3.0), g(1, NC, T, 2, sc=2.0)) for _ in range(2)]

    def plan(i, impl="auto"):
        e, n, c = inputs[i & 1]
        return ops.gru_plan(e, n, c, w_ih, w_hh, b_ih, b_hh, mlp_w, mlp_b, 5, 3, 4.0, 192.0, impl=impl)

    want = [plan(0, "steps").clone(), plan(1, "steps").clone()]
    hog = ops.ConvLayer(torch.randn(64, 384, 7, 7) / (384 * 49) ** 0.5, stride=2, padding=(3, 3), relu_post=True,
                        precision=_lib.CONV_BF16X6, device=dev)
    hog_x = torch.randn(15, 384, 96, 96, device=dev)
    s_hog, s_plan = torch.cuda.Stream(), torch.cuda.Stream()
    with torch.cuda.stream(s_hog):
        hog(hog_x)
    torch.cuda.synchronize()
    graphs = []
    if graph:
        for i in range(2):
            with torch.cuda.stream(s_plan):
                plan(i)
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=s_plan):
                o = plan(i)
            graphs.append((gr, o))
        torch.cuda.synchronize()
    outs = []
    for i in range(N):
        with torch.cuda.stream(s_hog):
            hog(hog_x)
        with torch.cuda.stream(s_plan):
            if graph:
                graphs[i & 1][0].replay()
                outs.append(graphs[i & 1][1].clone())
            else:
                outs.append(plan(i))
    torch.cuda.synchronize()
    wrong = [(i, float((o - want[i & 1]).abs().max())) for i, o in enumerate(outs) if not torch.equal(o, want[i & 1])]
    assert not wrong, f"{len(wrong)} of {N} persistent plan launches differ from the step path: {wrong[:5]}"
    diag = ops.gru_plan_diag(1, H, NC, 3, dev, stream=s_plan)
    assert diag["aborted_launches"] == 0 and diag["launches"] >= N, diag
