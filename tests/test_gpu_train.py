"""Training path on the GPU: the liblav_amd training ops (lav_pillar_decorate, lav_scatter_max + backward; through the
C ABI) against the reference goldens / a torch restatement, train-mode PointPillarNet gradients, and two optimisation
steps of train_lidar / train_bev against the REFERENCE trainers' loss terms (tests/golden/train.npz)."""
import os

import numpy as np
import pytest
import torch

import lav_amd
from lav_amd import ops
from lav_amd.train import LAV, TrainConfig, synthetic_bev_batch, synthetic_lidar_batch
from tests.test_oracle_golden import pillar_cases
from tests.util import CFG, state_dicts, sub_sd

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda")


@pytest.mark.parametrize("name", ["lidar", "uniform", "edge", "one_cell", "batch2"])
def test_pillar_decorate_vs_reference_golden(golden, name):
    g = golden["pillar"]
    clouds, n = pillar_cases(g)[name]
    n = n or [len(c) for c in clouds]
    ppn = lav_amd.PointPillarNet(16, [64, 64], **CFG)
    pts, n = ppn._pack([torch.from_numpy(c).to(DEV) for c in clouds], n)
    dec, uc, inv, src = ops.pillar_decorate(pts, n, ppn._grid)
    np.testing.assert_array_equal(uc.cpu().numpy(), g[f"{name}/unique_coords"])       # bit-exact indices
    np.testing.assert_array_equal(inv.cpu().numpy(), g[f"{name}/inverse"])
    ref = g[f"{name}/decorated"]
    got = dec.cpu().numpy()
    np.testing.assert_array_equal(got[:, :11], ref[:, :11])
    np.testing.assert_array_equal(got[:, 14:], ref[:, 14:])                            # cell-origin offsets: exact
    np.testing.assert_allclose(got[:, 11:14], ref[:, 11:14], rtol=0, atol=2e-5)        # float32 running mean vs fixed point
    flat = pts.reshape(-1, pts.shape[-1])[src.long()]
    assert torch.equal(flat, dec[:, :11]) and bool((src[1:] > src[:-1]).all())          # kept points stay in input order


def _ref_scatter_max(src, index, S):
    out = torch.full((S, src.shape[1]), float("-inf"), dtype=src.dtype, device=src.device)
    out = out.index_reduce(0, index, src, "amax", include_self=True)
    return out


@pytest.mark.parametrize("n,ch,S,seed", [(5000, 64, 700, 0), (333, 7, 50, 1), (64, 64, 1, 2)])
def test_scatter_max_forward_backward(n, ch, S, seed):
    g = torch.Generator().manual_seed(seed)
    src = torch.randn((n, ch), generator=g)
    src[::7] = src[::7].abs().floor()                      # ties inside segments
    index = torch.randint(0, S, (n,), generator=g)
    index[:S] = torch.arange(S)                            # every segment non-empty
    x = src.to(DEV).requires_grad_(True)
    idx = index.to(DEV).int()
    out, arg = ops.scatter_max(x, idx, S)
    xr = src.clone().requires_grad_(True)
    ref = _ref_scatter_max(xr, index, S)
    assert torch.equal(out.detach().cpu(), ref.detach())
    a = arg.cpu().long()
    assert torch.equal(src[a, torch.arange(ch)[None].expand(S, -1)], ref.detach()) and bool((index[a] == torch.arange(S)[:, None]).all())
    w = torch.randn((S, ch), generator=g)
    (out * w.to(DEV)).sum().backward()
    # reference gradient: all of grad_out goes to the arg-max row (ties: lowest row - torch's amax would split them)
    expect = torch.zeros_like(src)
    expect[a, torch.arange(ch)[None].expand(S, -1)] = w
    assert torch.equal(x.grad.cpu(), expect)


def test_scatter_max_empty_segment_convention():
    src = torch.tensor([[1.0, -2.0], [3.0, -5.0]], device=DEV)
    out, arg = ops.scatter_max(src, torch.tensor([2, 2], device=DEV, dtype=torch.int32), 4)
    assert out.cpu().tolist() == [[0, 0], [0, 0], [3, -2], [0, 0]] and arg.cpu().tolist() == [[2, 2], [2, 2], [1, 0], [2, 2]]


def test_pointpillar_train_mode_forward_and_gradients_vs_torch_restatement(golden):
    """Train-mode PointPillarNet (BatchNorm1d on batch statistics) on the HIP training ops == the reference's forward
    (point_pillar.py:92-116) restated with torch ops on the golden's decorated points, values and weight gradients."""
    g = golden["pillar"]
    clouds, n = pillar_cases(g)["batch2"]
    lsd, _ = state_dicts()
    ppn = lav_amd.PointPillarNet(16, [64, 64], **CFG)
    ppn.load_state_dict(sub_sd(lsd, "point_pillar_net."))
    ppn.train().to(DEV)
    canvas = ppn([torch.from_numpy(c).to(DEV) for c in clouds], n or [len(c) for c in clouds])
    wgt = torch.randn(canvas.shape, generator=torch.Generator().manual_seed(3)).to(DEV)
    (canvas * wgt).sum().backward()
    ref_net = lav_amd.PointPillarNet(16, [64, 64], **CFG)
    ref_net.load_state_dict(sub_sd(lsd, "point_pillar_net."))
    ref_net.train()
    dec = torch.from_numpy(g["batch2/decorated"])
    inv = torch.from_numpy(g["batch2/inverse"]).long()
    uc = torch.from_numpy(g["batch2/unique_coords"]).long()
    feat = ref_net.point_net.net(dec)
    fmax = _ref_scatter_max(feat, inv, len(uc))
    ref = torch.zeros((2, 64, 320, 320))
    ref[uc[:, 0], :, torch.clamp(319 - uc[:, 1], 0, 319), torch.clamp(uc[:, 2], 0, 319)] = fmax
    (ref * wgt.cpu()).sum().backward()
    np.testing.assert_allclose(canvas.detach().cpu().numpy(), ref.detach().numpy(), rtol=1e-4, atol=2e-5)
    for (k, p), (_, q) in zip(ppn.named_parameters(), ref_net.named_parameters()):
        # (a Linear bias in front of a train-mode BatchNorm has a mathematically zero gradient: both sides are noise)
        np.testing.assert_allclose(p.grad.cpu().numpy(), q.grad.numpy(), rtol=2e-3, atol=max(2e-3 * float(q.grad.abs().max()), 2e-4), err_msg=k)
    # BatchNorm running statistics moved identically
    np.testing.assert_allclose(ppn.point_net.net[1].running_mean.cpu().numpy(), ref_net.point_net.net[1].running_mean.numpy(), rtol=1e-4, atol=1e-5)


def test_train_bev_on_gpu_matches_reference_trainer(golden):
    ref = golden["train"]["bev_terms"]
    lav = LAV(TrainConfig(), DEV, what="bev")
    batch = synthetic_bev_batch(2, seed=11, num_objs=3)
    keys = ("plan_loss", "ego_cast_loss", "other_cast_loss", "cmd_loss")
    for step in range(2):
        torch.manual_seed(100 + step)
        info = lav.train_bev(*batch, other_weight=0.5)
        np.testing.assert_allclose([info[k] for k in keys], ref[step], rtol=1e-3 if step == 0 else 3e-2, atol=1e-4, err_msg=f"step {step}")


def test_train_lidar_on_gpu_matches_reference_trainer(golden):
    """Full train_full_v2 step (LiDARModel + UniPlanner distilled from the frozen BEVPlanner, detection + segmentation +
    motion losses, Adam) vs the reference's own LAV.train_lidar run on CPU with the same seeded weights and batch."""
    ref = golden["train"]["lidar_terms"]
    lav = LAV(TrainConfig(log_every=1), DEV, what="lidar")      # log_every=1: the log-only eval inference runs on every step, as in the reference
    batch = synthetic_lidar_batch(2, seed=12, max_points=20000, num_objs=3)
    keys = ("hm_loss", "box_loss", "ori_loss", "seg_loss", "plan_loss", "ego_cast_loss", "other_cast_loss", "cmd_loss")
    for step in range(2):
        torch.manual_seed(200 + step)
        info = lav.train_lidar(*batch)
        np.testing.assert_allclose([info[k] for k in keys], ref[step], rtol=2e-3 if step == 0 else 5e-2, atol=1e-4, err_msg=f"step {step}")
        assert info["ego_plan_locs"].shape == (20, 2) and np.isfinite(info["ego_plan_locs"]).all()


@pytest.mark.parametrize("general", [False, True])
@pytest.mark.parametrize("H,W,crop", [(40, 40, 24), (40, 40, 33), (56, 56, 24)])   # the reference only crops square maps
def test_crop_rotate_indexed_forward_backward_vs_grid_sample(H, W, crop, general, monkeypatch):
    """lav_crop_rotate_indexed / lav_crop_rotate_backward (crops of per-sample maps by index; the backward is a gather over the
    map's pixels: no atomics, bit-reproducible) vs torch's affine_grid + grid_sample on the materialised maps
    (uniplanner.py:310-352)."""
    from lav_amd.planner_common import crop_feature_torch
    if general:   # the kernel for geometries whose pre-image boxes do not fit the LDS stage (same arithmetic, gathers from L2)
        monkeypatch.setenv("LAV_CROP_BWD_GENERAL", "1")
    g = torch.Generator().manual_seed(4)
    feat = torch.randn((3, 40, H, W), generator=g)            # 40 channels: one full and one ragged channel block
    idx = torch.tensor([2, 0, 0, 1, 2], dtype=torch.int32)
    locs = torch.tensor([[0.0, 0.0], [3.0, -6.0], [-4.0, 2.0], [8.0, 8.0], [30.0, -30.0]])          # the last one mostly off the map
    oris = torch.tensor([0.0, 0.4, -1.1, 3.0, 0.2])
    f_gpu = feat.to(DEV).requires_grad_(True)
    out = ops.crop_rotate_indexed(f_gpu, idx.to(DEV), locs.to(DEV), oris.to(DEV), 2.0, crop, 0.0, 0.75)
    f_ref = feat.clone().requires_grad_(True)
    ref = crop_feature_torch(f_ref[idx.long()], locs, oris, 2.0, crop, 0.0, 0.75)
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().numpy(), rtol=0, atol=2e-5)
    w = torch.randn(ref.shape, generator=g)
    (out * w.to(DEV)).sum().backward()
    (ref * w).sum().backward()
    np.testing.assert_allclose(f_gpu.grad.cpu().numpy(), f_ref.grad.numpy(), rtol=0, atol=5e-5 * float(f_ref.grad.abs().max()))
    assert float(f_ref.grad[0].abs().sum()) > 0 and float(f_ref.grad[2].abs().sum()) > 0      # two crops share map 0: gradients accumulate
    first = f_gpu.grad.clone()
    f_gpu.grad = None
    out2 = ops.crop_rotate_indexed(f_gpu, idx.to(DEV), locs.to(DEV), oris.to(DEV), 2.0, crop, 0.0, 0.75)
    (out2 * w.to(DEV)).sum().backward()
    assert torch.equal(first, f_gpu.grad)                                                       # run-to-run identical bits


@pytest.mark.parametrize("R,T,I,H,shared", [(37, 7, 4, 512, False), (5, 20, 4, 48, False), (70, 20, 512, 384, True), (16, 1, 8, 64, False)])
def test_gru_seq_forward_backward_vs_torch_gru(R, T, I, H, shared):
    """lav_gru_seq_forward / _backward (one launch per step: MFMA recurrent GEMM + gates; gather-style backward, no atomics)
    against torch's nn.GRU on the CPU in fp32: outputs and the gradients of the input, the initial state and all four
    parameter tensors within 1e-4 of the largest reference value (uniplanner.py:255-308 in train mode).  `shared`: the
    same input at every step (the cast decoders) - projected once, its gradient summed over the steps."""
    g = torch.Generator().manual_seed(R * 1000 + H)
    gru = torch.nn.GRU(I, H, batch_first=True)
    with torch.no_grad():
        for p_ in gru.parameters():
            p_.copy_(torch.randn(p_.shape, generator=g) * (0.5 / H ** 0.5))
    u = torch.randn((R, I) if shared else (R, T, I), generator=g)
    h0 = torch.randn((R, H), generator=g) * 0.5
    w = torch.randn((R, T, H), generator=g)
    # reference
    u_ref, h0_ref = u.clone().requires_grad_(True), h0.clone().requires_grad_(True)
    seq = u_ref[:, None].expand(-1, T, -1).contiguous() if shared else u_ref
    out_ref, _ = gru(seq, h0_ref[None])
    (out_ref * w).sum().backward()
    ref_grads = [u_ref.grad, h0_ref.grad] + [p_.grad.clone() for p_ in gru.parameters()]
    # HIP
    par = [p_.detach().clone().to(DEV).requires_grad_(True) for p_ in gru.parameters()]        # w_ih, w_hh, b_ih, b_hh
    u_gpu, h0_gpu = u.to(DEV).requires_grad_(True), h0.to(DEV).requires_grad_(True)
    out = ops.gru_seq(torch.nn.functional.linear(u_gpu, par[0], par[2]), h0_gpu, par[1], par[3], T)
    (out * w.to(DEV)).sum().backward()
    scale = float(out_ref.detach().abs().max())
    np.testing.assert_allclose(out.detach().cpu().numpy(), out_ref.detach().numpy(), rtol=0, atol=1e-4 * scale)
    for name, got, want in zip(("du", "dh0", "dw_ih", "dw_hh", "db_ih", "db_hh"), [u_gpu.grad, h0_gpu.grad] + [p_.grad for p_ in par], ref_grads):
        np.testing.assert_allclose(got.cpu().numpy(), want.numpy(), rtol=0, atol=1e-4 * float(want.abs().max()), err_msg=name)
    # bit-reproducible (no atomics anywhere in the layer's own kernels)
    first = h0_gpu.grad.clone()
    h0_gpu.grad = None
    out2 = ops.gru_seq(torch.nn.functional.linear(u_gpu, par[0], par[2]), h0_gpu, par[1], par[3], T)
    (out2 * w.to(DEV)).sum().backward()
    assert torch.equal(out, out2) and torch.equal(first, h0_gpu.grad)


def test_train_lidar_loss_curve_vs_reference_trainer(golden):
    """BASELINE.json config #5 ("loss-curve match to reference for 500 steps"): the reference's LAV.train_lidar
    (lav/lav_final_v2.py:140-259, the loop of lav/train_full_v2.py:24-46) ran 500 optimisation steps on CPU over four
    alternating seeded batches (tests/golden/make_golden.py:gold_train_curve, tests/golden/train_curve.npz); the MI355X
    trainer runs the same 500 steps from the same weights.

    What "match" can mean: step 0 is the same arithmetic (2e-3).  From then on two float32 implementations separate -
    Adam's first updates are lr*sign(g), so rounding-level gradient differences (oneDNN's convolutions on the reference's CPU
    run, MIOpen's here) flip signs where g ~ 0 - and the trajectories decorrelate like any chaotic system's.  The MI355X
    trainer itself IS reproducible: every liblav_amd kernel reduces in a fixed order, and with torch / MIOpen switched to their
    deterministic algorithms (lav_amd.train.run.set_deterministic, as this test does) two runs agree bit for bit
    (test_train_lidar_step_is_bit_reproducible_with_deterministic_algorithms; without the switch torch's atomics-based
    backward kernels and MIOpen's algorithm choice make two runs differ by up to 29 % in the 100-step moving average - measured,
    tools/determinism_probe.py, tools/curve_run.py).  That reproducibility is per PROCESS: which convolution algorithms MIOpen
    picks depends on what ran before in the process (its find results are cached), so this test's curve is a fixed function of
    the test session, not of the seeds alone - measured as the only GPU test of a process: first 60 smoothed steps within 1.5 %
    of the reference's, 100-step moving average within 29.4 %, final level 6.2 against the reference's 8.7; at the end of the
    full GPU suite: 1.7 %, 35.3 %, 5.7 (the reference's single CPU run is one more sample of that spread; MI355X runs tend to
    end lower).  The bars (round 5): the envelope of eight CPU-port curves and the reference, see below - on the curve's shape, not on
    per-step values."""
    import subprocess
    import sys
    import tempfile
    ref = golden["train_curve"]["terms"]                       # (steps, 8)
    keys = [str(k) for k in golden["train_curve"]["keys"]]
    steps = len(ref)
    assert steps == 500
    # Round 6: the 500 steps run in a process of their own with an empty MIOpen database (tests/_curve_worker.py) and with MIOpen in
    # immediate mode (set_deterministic: no timed solver search) - the curve is then ONE fixed curve of the code (two runs: the same 500
    # rows), not a sample of what the solver timings of a process happened to pick (see the worker's header; measured this round:
    # 50.1 -> 5.8, first 60 smoothed steps within 1.5 %, every term inside the envelope).
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "curve.npz")
        worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_curve_worker.py")
        r = subprocess.run([sys.executable, worker, out, ",".join(keys), str(steps)], capture_output=True, text=True, timeout=1500)
        assert r.returncode == 0, f"curve worker failed:\n{r.stdout[-2000:]}\n{r.stderr[-4000:]}"
        rows = np.load(out)["rows"]
    ours = np.array(rows)
    assert np.isfinite(ours).all()
    np.testing.assert_allclose(ours[0], ref[0], rtol=2e-3, atol=1e-4, err_msg="step 0")
    np.testing.assert_allclose(ours[:4].sum(1), ref[:4].sum(1), rtol=0.15, err_msg="first pass over the four batches")
    smooth = lambda a, w: np.convolve(a, np.ones(w) / w, mode="valid")
    tot_o, tot_r = ours.sum(1), ref.sum(1)
    early = np.abs(smooth(tot_o, 25)[:60] - smooth(tot_r, 25)[:60]) / smooth(tot_r, 25)[:60]
    whole = np.abs(smooth(tot_o, 100) - smooth(tot_r, 100)) / smooth(tot_r, 100)
    final_o, final_r = tot_o[-100:].mean(), tot_r[-100:].mean()
    print(f"loss curve over {steps} steps: reference {smooth(tot_r, 25)[0]:.1f} -> {final_r:.1f}, MI355X {smooth(tot_o, 25)[0]:.1f} -> {final_o:.1f}; "
          f"deviation of the smoothed total: first 60 steps {early.max():.3f}, 100-step average {whole.max():.3f}")
    assert final_r < 0.3 * smooth(tot_r, 25)[0], "the reference run must actually learn for the comparison to mean something"
    # Round 5: bars from an ENVELOPE instead of bars widened to fit.  tests/golden/train_curve_envelope.npz holds the
    # CPU port of this very trainer (torch CPU ops, tools/curve_cpu.py) run with 1 ... 8 threads: eight more float32 implementations
    # of the same 500 steps.  Against the reference's single run they end at 0.64x ... 1.08x of its final level (5 and 8 threads: 0.64x -
    # the level every MI355X session ends near: 0.63-0.75x, which round 4 had taken for a bias; 2 / 4 / 6 / 7 threads: 0.92 / 0.87 / 0.89 /
    # 0.99x), leave its 100-step average by 17-103 % and its first 60 smoothed steps by 1.2-5.4 % (3 threads: 5.4 %; the one MI355X
    # session at 5.7 % was such a sample too).
    # The MI355X curve is held to what those samples and the reference span - as far as the family holds itself to it (below).
    env = golden["train_curve_envelope"]["curves"]                          # (samples, 500, 8)
    assert [str(k) for k in golden["train_curve_envelope"]["keys"]] == keys and env.shape[0] >= 6 and env.shape[1:] == ref.shape
    samples = np.concatenate([env, ref[None]], axis=0)

    def excess(curve, family):
        """How far the 100-step moving average of `curve` leaves [min, max] of the family's, widened by 10 % (+0.002 for terms that have
        fallen to ~0.01): the largest excursion over the run, <= 0 inside."""
        st = np.stack([smooth(c, 100) for c in family])
        o = smooth(curve, 100)
        return float(np.maximum(st.min(0) * 0.9 - 0.002 - o, o - st.max(0) * 1.1 - 0.002).max())

    # The family is heavy-tailed: left out in turn, the samples THEMSELVES leave the envelope of the other eight - the reference's
    # own box loss by 0.37, the 3-thread port's plan loss by 9.9 (the one event of the run, batch 0's plan loss leaving its plateau,
    # happens at a different step) - so "inside the min-max of nine" is not a test a family member passes.  The bar per term comes
    # from the family's own worst leave-one-out excursion E: a tenth member of an exchangeable family exceeds the worst of nine with
    # probability 1/10 per term (one session measured 0.0209 on the orientation loss against E = 0.0173), and nine correlated terms
    # are tested, so the bar is 2 E: the MI355X curve may leave the envelope of all nine by no more than twice what the worst of them
    # leaves the envelope of the other eight.  For the terms whose samples all stay inside (heat map, segmentation, command) E = 0:
    # inside the 10 % envelope at every step.
    series = {k: (lambda c, j=j: c[:, j]) for j, k in enumerate(keys)}
    series["total"] = lambda c: c.sum(1)
    report = {}
    for k, f in series.items():
        fam = [f(c) for c in samples]
        bar = 2.0 * max(0.0, max(excess(fam[i], fam[:i] + fam[i + 1:]) for i in range(len(fam))))
        got = excess(f(ours), fam)
        report[k] = (round(got, 4), round(bar, 4))
        assert got <= bar + 1e-9, f"{k}: the 100-step average leaves the envelope of the nine samples by {got:.4f}; bar (twice the samples' worst leave-one-out excursion) {bar:.4f}"
    print("excursion beyond the envelope (MI355X, bar = 2 x worst family member):", report)
    finals = samples[:, -100:, :].sum(2).mean(1)
    assert 0.9 * finals.min() <= final_o <= 1.1 * finals.max(), f"final level {final_o:.2f} outside the samples' {finals.min():.2f} .. {finals.max():.2f}"
    early_env = max((np.abs(smooth(c.sum(1), 25)[:60] - smooth(tot_r, 25)[:60]) / smooth(tot_r, 25)[:60]).max() for c in env)
    assert early.max() <= 1.1 * early_env, f"the first 60 smoothed steps leave the reference's by {early.max():.3f} (CPU samples: up to {early_env:.3f})"
    # and the reproducible terms follow the reference itself closely (every run measured, on the MI355X and on the CPU, within 2 %)
    for k, bar in (("hm_loss", 0.05), ("seg_loss", 0.05), ("cmd_loss", 0.05)):
        j = keys.index(k)
        dev = (np.abs(smooth(ours[:, j], 100) - smooth(ref[:, j], 100)) / smooth(ref[:, j], 100)).max()
        assert dev < bar, f"{k}: 100-step moving average leaves the reference's by {dev:.3f} (bar {bar})"
    # the loss terms that training drives down go down here too (detection heat-map, box, orientation, motion terms)
    for j, k in enumerate(keys):
        if ref[-100:, j].mean() < 0.5 * ref[:20, j].mean():
            assert ours[-100:, j].mean() < 0.75 * ours[:20, j].mean(), f"{k} did not decrease like the reference's"


def test_train_lidar_loss_curve_batch8_config5_clouds(golden):
    """Config #5 at its own cloud size: the reference trainer on CPU, batch 8 x 120 000-point clouds, two alternating seeded batches,
    100 steps (tests/golden/make_golden.py train_curve_b8 - the largest batch its CPU run finishes in the build container; batch 32
    would be four times that) against the MI355X trainer on the same batches.  Over these 100 steps the run is still in the
    reproducible regime (profiles/r04_loss_curve_bisect.md: all runs agree for ~100 steps): every loss term's 20-step average must
    follow the reference's (5 % on the perception terms, 25 % on the planner terms, 20 % on the total)."""
    from lav_amd.train.run import set_deterministic
    g = golden["train_curve_b8"]
    ref, keys = g["terms"], [str(k) for k in g["keys"]]
    B, pts, nb, seed0 = int(g["batch"][0]), int(g["points"][0]), int(g["nbatches"][0]), int(g["seed0"][0])
    assert (B, pts) == (8, 120000) and len(ref) >= 100
    set_deterministic(True)
    try:
        torch.manual_seed(0)
        lav = LAV(TrainConfig(log_inference=False), DEV, what="lidar")
        batches = [synthetic_lidar_batch(B, seed=seed0 + i, max_points=pts, num_objs=3) for i in range(nb)]
        rows = []
        for step in range(len(ref)):
            torch.manual_seed(1000 + step)
            info = lav.train_lidar(*batches[step % nb])
            rows.append([info[k] for k in keys])
    finally:
        set_deterministic(False)
    ours = np.array(rows)
    assert np.isfinite(ours).all()
    np.testing.assert_allclose(ours[0], ref[0], rtol=2e-3, atol=1e-4, err_msg="step 0")
    smooth = lambda a, w: np.convolve(a, np.ones(w) / w, mode="valid")
    tot_o, tot_r = ours.sum(1), ref.sum(1)
    dev_tot = (np.abs(smooth(tot_o, 20) - smooth(tot_r, 20)) / smooth(tot_r, 20)).max()
    devs = {k: float((np.abs(smooth(ours[:, j], 20) - smooth(ref[:, j], 20)) / smooth(ref[:, j], 20)).max()) for j, k in enumerate(keys)}
    print(f"batch-8 curve over {len(ref)} steps: reference {tot_r[:4].mean():.1f} -> {tot_r[-20:].mean():.1f}, MI355X {tot_o[:4].mean():.1f} -> {tot_o[-20:].mean():.1f}; "
          f"max deviation of the 20-step average: total {dev_tot:.3f}, per term {({k: round(v, 3) for k, v in devs.items()})}")
    assert tot_r[-20:].mean() < 0.5 * tot_r[:4].mean(), "the reference run must learn"
    # measured in two sessions: 0.053 and - after the frozen teacher's crop stems moved to the tap-pair split kernel, which shifts its
    # waypoints by ~1e-5 - 0.120 (plan / cast terms 0.15-0.17, hm / seg / cmd 0.003-0.012): the planner terms already carry the
    # sensitivity that profiles/r04_loss_curve_bisect.md shows over 500 steps, the perception terms do not
    assert dev_tot < 0.20, f"total loss leaves the reference's 20-step average by {dev_tot:.3f}"
    for k in ("hm_loss", "seg_loss", "cmd_loss"):
        assert devs[k] < 0.05, (k, devs[k])
    for k in ("plan_loss", "ego_cast_loss", "other_cast_loss", "box_loss", "ori_loss"):   # measured 0.05-0.12
        assert devs[k] < 0.30, (k, devs[k])


def test_train_lidar_loss_curve_batch32_config5_as_stated(golden):
    """BASELINE.json config #5 AS STATED - batch 32 x 120 000-point clouds (round 5): 25 steps of the reference trainer on CPU
    (tests/golden/make_golden.py train_curve_b32, about a minute per step in the build container) against the MI355X trainer on the
    same batch and weights.  Step 0 is the same arithmetic (2e-3); over 25 steps every term's 5-step average follows the
    reference's (the bars of the batch-8 test: 5 % perception, 30 % planner terms, 20 % total)."""
    from lav_amd.train.run import set_deterministic
    g = golden["train_curve_b32"]
    ref, keys = g["terms"], [str(k) for k in g["keys"]]
    B, pts, nb, seed0 = int(g["batch"][0]), int(g["points"][0]), int(g["nbatches"][0]), int(g["seed0"][0])
    assert (B, pts) == (32, 120000) and len(ref) >= 20
    set_deterministic(True)
    try:
        torch.manual_seed(0)
        lav = LAV(TrainConfig(log_inference=False), DEV, what="lidar")
        batches = [synthetic_lidar_batch(B, seed=seed0 + i, max_points=pts, num_objs=3) for i in range(nb)]
        rows = []
        for step in range(len(ref)):
            torch.manual_seed(1000 + step)
            info = lav.train_lidar(*batches[step % nb])
            rows.append([info[k] for k in keys])
    finally:
        set_deterministic(False)
    ours = np.array(rows)
    assert np.isfinite(ours).all()
    np.testing.assert_allclose(ours[0], ref[0], rtol=2e-3, atol=1e-4, err_msg="step 0")
    smooth = lambda a, w: np.convolve(a, np.ones(w) / w, mode="valid")
    tot_o, tot_r = ours.sum(1), ref.sum(1)
    dev_tot = (np.abs(smooth(tot_o, 5) - smooth(tot_r, 5)) / smooth(tot_r, 5)).max()
    devs = {k: float((np.abs(smooth(ours[:, j], 5) - smooth(ref[:, j], 5)) / smooth(ref[:, j], 5)).max()) for j, k in enumerate(keys)}
    print(f"batch-32 curve over {len(ref)} steps: reference {tot_r[0]:.1f} -> {tot_r[-5:].mean():.1f}, MI355X {tot_o[0]:.1f} -> {tot_o[-5:].mean():.1f}; "
          f"max deviation of the 5-step average: total {dev_tot:.3f}, per term {({k: round(v, 3) for k, v in devs.items()})}")
    assert tot_r[-5:].mean() < 0.8 * tot_r[0], "the reference run must learn"
    assert dev_tot < 0.20, f"total loss leaves the reference's 5-step average by {dev_tot:.3f}"
    for k in ("hm_loss", "seg_loss", "cmd_loss"):
        assert devs[k] < 0.05, (k, devs[k])
    for k in ("plan_loss", "ego_cast_loss", "other_cast_loss", "box_loss", "ori_loss"):
        assert devs[k] < 0.30, (k, devs[k])


def test_train_full_driver_reads_recorded_routes(tmp_path):
    """train_full_v2.py without --synthetic on the MI355X: one epoch over a 3-frame synthetic route through the
    'temporal_lidar_painted' loader of lav_amd.data (one batch of 2, drop_last) - loader dtypes / shapes meet the HIP training
    step, checkpoints are written under the reference's names."""
    import subprocess
    import sys
    from tests.util import dataset_fixture_config
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = dataset_fixture_config(str(tmp_path), routes=1, frames=23)
    base = [sys.executable, os.path.join(repo, "train_full_v2.py"), "--config-path", cfg, "--batch-size", "2", "--num-epoch", "1",
            "--num-workers", "0", "--save-dir", str(tmp_path / "ck")]
    # a run on recorded routes follows the reference's checkpoint rules (lav_final_v2.py:42-72): the teacher, the LiDAR model and
    # the planner come from the config's *_model_dir keys or the command line - never silently from seeded random weights
    out = subprocess.run(base, capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and "bev_model_dir" in (out.stdout + out.stderr)
    import torch
    from lav_amd.train import LAV, TrainConfig
    seeded = LAV(TrainConfig(), torch.device("cpu"), what="lidar")
    names = dict(bev=seeded.bev_planner.state_dict(), lidar=seeded.state_dict("lidar"), uniplanner=seeded.state_dict("uniplanner"))
    for k, sd in names.items():
        torch.save(sd, tmp_path / f"{k}_seed.th")
    out = subprocess.run(base + ["--bev", str(tmp_path / "bev_seed.th"), "--lidar", str(tmp_path / "lidar_seed.th"),
                                 "--uniplanner", str(tmp_path / "uniplanner_seed.th")], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert '"steps": 1' in out.stdout and "3 recorded frames" in out.stdout
    assert os.path.exists(tmp_path / "ck" / "lidar_1.th") and os.path.exists(tmp_path / "ck" / "uniplanner_1.th")


def test_two_rank_data_parallel_step_over_the_hip_autograd_functions(tmp_path):
    """Two ranks of train_full_v2.py on this one GPU (gloo moves the gradients through the host - RCCL refuses two ranks on
    a device; on a multi-GPU node the same code runs over RCCL): DistributedDataParallel's gradient hooks must fire for the
    parameters that reach the loss through liblav_amd's autograd functions (sequence GRU, indexed crops, pillar ops) and
    through find_unused_parameters for those that do not; after two optimisation steps both replicas hold identical weights."""
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", "29641", os.path.join(repo, "train_full_v2.py"), "--synthetic", "--batch-size", "2", "--num-epoch", "1",
                          "--steps-per-epoch", "2", "--max-points", "20000", "--save-dir", str(tmp_path / "ck")],
                         capture_output=True, text=True, timeout=900, env=dict(os.environ, LAV_DIST_BACKEND="gloo"))
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    assert '"n_gpus": 2' in out.stdout and '"replicas_in_sync": true' in out.stdout and '"steps": 2' in out.stdout, out.stdout[-1500:]


@pytest.mark.parametrize("shape", [(4, 64, 40, 40), (3, 5, 7, 9), (57, 512, 3, 3)])
@pytest.mark.parametrize("mode", ["plain", "relu_pre", "relu_post", "residual"])
def test_bn_act_forward_backward_vs_torch(shape, mode):
    """lav_bn_train_forward / _backward (batch statistics + fused ReLU / residual) against nn.BatchNorm2d + F.relu in float64 on the
    host: output, running statistics and all four gradients; and bit-reproducible from run to run."""
    from lav_amd.train.hipnn import bn_act
    torch.manual_seed(3)
    dev = torch.device("cuda")
    B, C, H, W = shape
    x = torch.randn(shape) * 1.5 + 0.3
    res = torch.randn(shape) if mode == "residual" else None
    dy = torch.randn(shape)
    kw = dict(relu_pre=mode == "relu_pre", relu_post=mode in ("relu_post", "residual"))

    def run(device, dtype, force_torch=False):
        bn = torch.nn.BatchNorm2d(C, eps=1e-3, momentum=0.01).to(device=device, dtype=dtype).train()
        with torch.no_grad():
            bn.weight.copy_(torch.linspace(0.5, 1.5, C)); bn.bias.copy_(torch.linspace(-0.3, 0.3, C))
        xx = x.to(device=device, dtype=dtype).requires_grad_(True)
        rr = None if res is None else res.to(device=device, dtype=dtype).requires_grad_(True)
        y = bn_act(bn, xx, residual=rr, **kw)
        y.backward(dy.to(device=device, dtype=dtype))
        out = [y.detach(), bn.running_mean, bn.running_var, xx.grad, bn.weight.grad, bn.bias.grad]
        if rr is not None:
            out.append(rr.grad)
        return [t.detach().double().cpu() for t in out], int(bn.num_batches_tracked)

    ref, nb_ref = run("cpu", torch.float64)
    got, nb = run(dev, torch.float32)
    again, _ = run(dev, torch.float32)
    assert nb == nb_ref == 1
    names = ["y", "running_mean", "running_var", "dx", "dgamma", "dbeta", "dres"]
    for n, a, b, c in zip(names, got, ref, again):
        assert torch.equal(a, c), f"{n}: two runs differ"
        scale = max(b.abs().max().item(), 1.0)
        assert (a - b).abs().max().item() <= 1e-4 * scale, (n, (a - b).abs().max().item(), scale)


@pytest.mark.parametrize("mode", ["relu_pre", "relu_post", "residual"])
def test_bn_act_leaves_the_bounds_of_what_it_writes(mode):
    """lav_bn_train_forward_amax / _backward_amax (round 6): under hipnn.use_precision("f16x3") the fused BatchNorm tags its output and
    its input gradient with the per-workgroup maxima of what the launch wrote - exactly max |y| resp. max |dx| -, the convolution that
    reads the tensor takes them (no measuring launch), an in-place change voids the tag, and the values are those of the untagged run."""
    from lav_amd import ops
    from lav_amd.train import hipnn
    torch.manual_seed(7)
    dev = torch.device("cuda")
    B, C, H, W = 3, 64, 40, 48
    x = (torch.randn((B, C, H, W)) * 2.0).to(dev).requires_grad_(True)
    res = torch.randn((B, C, H, W)).to(dev) if mode == "residual" else None
    bn = torch.nn.BatchNorm2d(C, eps=1e-3).to(dev).train()
    conv = torch.nn.Conv2d(C, 64, 3, 1, 1, bias=False).to(dev)
    kw = dict(relu_pre=mode == "relu_pre", relu_post=mode in ("relu_post", "residual"))
    with hipnn.use_precision("f16x3"):
        y = hipnn.bn_act(bn, x, residual=res, **kw)
        am = hipnn._trusted(y)
        assert am is not None and am.count == _parts(B, C, H * W)
        assert am.buf[:am.count].max().item() == y.detach().abs().max().item()
        ops.train_work["absmax_launches"] = 0
        z = hipnn.conv_module(conv, y)
        assert ops.train_work["absmax_launches"] == 0, "the convolution measured a tensor that carried its bound"
        seen = {}
        y.register_hook(lambda g: seen.setdefault("am", hipnn._trusted(g)) and None)   # the gradient object the BatchNorm's backward will read
        z.square().sum().backward()
        gx = x.grad.clone()
        # the BatchNorm's input gradient left its own bound
        tag = hipnn._trusted(x.grad)
        if tag is not None:     # (leaf gradients are accumulated into .grad by autograd: the object may be a copy)
            assert tag.buf[:tag.count].max().item() == x.grad.abs().max().item()
        # an in-place change voids a tag
        y2 = hipnn.bn_act(bn, x.detach(), residual=res, **kw)
        assert hipnn._trusted(y2) is not None
        y2.mul_(2.0)
        assert hipnn._trusted(y2) is None
    # the same step without the tags (LAV_TRAIN_BN_AMAX=0): same bits
    os.environ["LAV_TRAIN_BN_AMAX"] = "0"
    try:
        bn2 = torch.nn.BatchNorm2d(C, eps=1e-3).to(dev).train()
        x2 = x.detach().clone().requires_grad_(True)
        with hipnn.use_precision("f16x3"):
            y3 = hipnn.bn_act(bn2, x2, residual=res, **kw)
            assert hipnn._trusted(y3) is None
            hipnn.conv_module(conv, y3).square().sum().backward()
        assert torch.equal(y3, y) and torch.equal(x2.grad, gx)
    finally:
        os.environ.pop("LAV_TRAIN_BN_AMAX")


def _parts(B, C, hw):
    from lav_amd import _lib
    return _lib.load().lav_bn_train_amax_count(B, C, hw)


def test_heads_train_mode_fused_first_convolution_matches_per_head_modules():
    """LiDARModel.heads in train mode (one 384 -> 256 convolution + one BatchNorm launch pair over the four heads) against the
    four Head modules run one by one through torch ops: outputs, feature gradient, every parameter gradient, running stats."""
    import copy
    torch.manual_seed(5)
    dev = torch.device("cuda")
    lm = lav_amd.LiDARModel(num_input=16, backbone="cnn", num_features=[64, 64], **CFG)
    lm.load_state_dict(state_dicts()[0])
    lm = lm.to(dev).train()
    ref = copy.deepcopy(lm)
    feats = torch.randn((2, 384, 40, 40), device=dev)
    fa, fb = feats.clone().requires_grad_(True), feats.clone().requires_grad_(True)
    outs = lm.heads(fa)
    outs_ref = tuple(torch.sigmoid(h.net(fb)) if h._sigmoid else h.net(fb) for h in (getattr(ref, n) for n in lm.ALL_HEADS))
    gs = [torch.randn_like(o) for o in outs]
    torch.autograd.backward(outs, gs)
    torch.autograd.backward(outs_ref, gs)
    for a, b in zip(outs, outs_ref):
        assert (a - b).abs().max().item() <= 1e-4 * max(1.0, b.abs().max().item())
    assert (fa.grad - fb.grad).abs().max().item() <= 1e-4 * fb.grad.abs().max().item()
    pa, pb = dict(lm.named_parameters()), dict(ref.named_parameters())
    for k, v in pb.items():
        if "_head" in k:
            assert (pa[k].grad - v.grad).abs().max().item() <= 2e-4 * max(v.grad.abs().max().item(), 1e-3), k
    ba, bb = dict(lm.named_buffers()), dict(ref.named_buffers())
    for k, v in bb.items():
        if "_head" in k:
            assert (ba[k].double() - v.double()).abs().max().item() <= 1e-5 * max(1.0, v.double().abs().max().item()), k


def test_train_lidar_step_is_bit_reproducible_with_deterministic_algorithms():
    """Two trainers built from the same seeds and driven over the same batches agree BIT FOR BIT - losses of every step and every
    parameter after the last - once torch / MIOpen are told to use deterministic algorithms (set_deterministic): liblav_amd's own
    forward and backward kernels (pillar decorate, scatter-max, crop gather, GRU tape, BatchNorm) reduce in a fixed order."""
    from lav_amd.train.run import set_deterministic
    batches = [synthetic_lidar_batch(2, seed=40 + i, max_points=20000, num_objs=3) for i in range(2)]

    def run():
        torch.manual_seed(0)
        lav = LAV(TrainConfig(log_inference=False), DEV, what="lidar")
        rows = []
        for s in range(6):
            torch.manual_seed(1000 + s)
            info = lav.train_lidar(*batches[s % 2])
            rows.append([v for v in info.values() if isinstance(v, float)])
        return np.array(rows), [p.detach().clone() for p in lav.student.parameters()]

    set_deterministic(True)
    try:
        (a, pa), (b, pb) = run(), run()
    finally:
        set_deterministic(False)
    assert np.array_equal(a, b), np.abs(a - b).max()
    assert all(torch.equal(x, y) for x, y in zip(pa, pb))


def test_cpu_baseline_leg_computes_the_same_step_as_the_gpu_trainer():
    """The CPU sample bench.py reports beside the train_full line (oracle stand-ins for the HIP-only pieces, torch CPU ops for the
    rest) is the same optimisation step: all loss terms of step 0 agree with the MI355X trainer's."""
    from lav_amd.point_pillar import PointPillarNet
    from oracle import train_cpu
    batch = synthetic_lidar_batch(2, seed=40, max_points=20000, num_objs=3)
    g = LAV(TrainConfig(log_inference=False), DEV, what="lidar")
    torch.manual_seed(1000)
    gpu = g.train_lidar(*batch)
    restore = PointPillarNet.forward_train
    PointPillarNet.forward_train = lambda self, lidars, num_points: train_cpu.pillar_forward_train(self, lidars, num_points)
    try:
        lav = LAV(TrainConfig(log_inference=False), torch.device("cpu"), what="lidar")
        torch.manual_seed(1000)
        with train_cpu.teacher_on_cpu(lav.bev_planner):
            cpu = lav.train_lidar(*batch)
    finally:
        PointPillarNet.forward_train = restore
    for k, v in gpu.items():
        if isinstance(v, float):
            assert abs(cpu[k] - v) <= 2e-3 * max(1.0, abs(v)), (k, cpu[k], v)


@pytest.mark.parametrize("B,cin,cout,H,W,S,KS", [(2, 64, 64, 20, 32, 1, 3), (3, 128, 64, 9, 40, 1, 3), (1, 64, 128, 33, 16, 1, 3), (2, 384, 256, 12, 48, 1, 3),
                                                 (5, 64, 64, 7, 4, 1, 3), (2, 64, 64, 40, 64, 2, 3), (3, 64, 128, 18, 80, 2, 3), (1, 128, 64, 66, 8, 2, 3),
                                                 (2, 128, 256, 6, 40, 2, 3), (2, 128, 64, 24, 40, 2, 7), (3, 64, 64, 10, 8, 2, 7), (1, 384, 64, 96, 96, 2, 7)])
def test_conv_wgrad_vs_float64(B, cin, cout, H, W, S, KS):
    """lav_conv_wgrad (round 5: 3x3 weight gradient of stride 1 / 2 on the bf16 matrix cores, operands split exactly into three bf16
    pieces) against the float64 weight gradient of torch's convolution on the CPU: ragged 16-pixel segments (W = 40, 4; stride 2:
    output widths 40, 4, 20), blocks of rows with their halo rows (H = 33, 66), several co / ci tiles, the heads' channel counts; the 7x7 stride-2 stem (one ky per task) at ragged and at its real size.  Bar: 2e-6 of sum |dy||x| per weight (what the forward
    split kernel is held to, tests/test_gpu_glue.py), and run-to-run bit equality (partial sums are added in a fixed order)."""
    import ctypes as C
    from lav_amd import _lib
    from lav_amd.ops import _ptr, _stream, _workspace, check
    g = torch.Generator().manual_seed(B * 1000 + cin + H)
    x = torch.randn((B, cin, H, W), generator=g)
    dy = torch.randn((B, cout, H // S, W // S), generator=g)
    ref = torch.nn.grad.conv2d_weight(x.double(), (cout, cin, KS, KS), dy.double(), stride=S, padding=KS // 2)
    mag = torch.nn.grad.conv2d_weight(x.double().abs(), (cout, cin, KS, KS), dy.double().abs(), stride=S, padding=KS // 2)
    lib = _lib.load()
    xd, dyd = x.to(DEV), dy.to(DEV)
    nbytes = lib.lav_conv_wgrad_workspace_bytes(B, cin, cout, H, W, KS, S)
    assert nbytes > 0
    ws = _workspace("conv_wgrad_test", nbytes, DEV)
    outs = []
    for _ in range(2):
        dw = torch.full((cout, cin, KS, KS), float("nan"), device=DEV)
        check(lib.lav_conv_wgrad(_ptr(xd), _ptr(dyd), B, cin, cout, H, W, KS, S, _ptr(dw), _ptr(ws), ws.numel(), _stream()), "lav_conv_wgrad")
        outs.append(dw.cpu())
    assert torch.equal(outs[0], outs[1]), "the weight gradient must be bit-reproducible"
    err = ((outs[0].double() - ref).abs() / mag.clamp_min(1e-30)).max().item()
    assert err < 2e-6, f"max |dw - ref| / sum|dy||x| = {err:.3e}"
    # round 6: the same gradient on two fp16 pieces per operand (lav_conv_wgrad_amax) from measured maxima - operands spanning five
    # decades of magnitude, the same bar
    gn = np.random.Generator(np.random.PCG64(B * 77 + cin + W))
    x5 = torch.from_numpy((gn.standard_normal(tuple(x.shape)) * np.exp(gn.uniform(-8.0, 3.0, tuple(x.shape)))).astype(np.float32))
    dy5 = torch.from_numpy((gn.standard_normal(tuple(dy.shape)) * np.exp(gn.uniform(-9.0, 2.0, tuple(dy.shape)))).astype(np.float32))
    ref5 = torch.nn.grad.conv2d_weight(x5.double(), (cout, cin, KS, KS), dy5.double(), stride=S, padding=KS // 2)
    mag5 = torch.nn.grad.conv2d_weight(x5.double().abs(), (cout, cin, KS, KS), dy5.double().abs(), stride=S, padding=KS // 2)
    x5d, dy5d = x5.to(DEV), dy5.to(DEV)
    ax, ay = torch.zeros(512, device=DEV), torch.zeros(512, device=DEV)
    check(lib.lav_absmax_parts(_ptr(x5d), x5d.numel(), _ptr(ax), _stream()), "lav_absmax_parts")
    check(lib.lav_absmax_parts(_ptr(dy5d), dy5d.numel(), _ptr(ay), _stream()), "lav_absmax_parts")
    assert ax.max().item() == x5.abs().max().item() and ay.max().item() == dy5.abs().max().item()
    outs = []
    for _ in range(2):
        dw = torch.full((cout, cin, KS, KS), float("nan"), device=DEV)
        check(lib.lav_conv_wgrad_amax(_ptr(x5d), _ptr(dy5d), B, cin, cout, H, W, KS, S, _ptr(dw), _ptr(ws), ws.numel(), _ptr(ax), 512, _ptr(ay), 512,
                                      _stream()), "lav_conv_wgrad_amax")
        outs.append(dw.cpu())
    assert torch.equal(outs[0], outs[1]), "the fp16-piece weight gradient must be bit-reproducible"
    err = ((outs[0].double() - ref5).abs() / mag5.clamp_min(1e-30)).max().item()
    assert err < 2e-6, f"f16x3: max |dw - ref| / sum|dy||x| = {err:.3e}"
    assert lib.lav_conv_wgrad_workspace_bytes(1, 48, 64, 8, 8, 3, 1) == 0       # 48 input channels: not a multiple of 64
    assert lib.lav_conv_wgrad_workspace_bytes(1, 64, 64, 8, 12, 3, 2) == 0      # stride 2: the width must be a multiple of 8
    assert lib.lav_conv_wgrad_workspace_bytes(1, 64, 64, 8, 8, 3, 3) == 0
    assert lib.lav_conv_wgrad_workspace_bytes(1, 64, 64, 8, 8, 7, 1) == 0       # 7x7: stride 2 only


@pytest.mark.parametrize("prec", ["bf16x6", "f16x3"])
@pytest.mark.parametrize("B,cin,cout,k,s,H,W", [(2, 64, 64, 3, 1, 24, 32), (2, 64, 128, 3, 2, 20, 24), (3, 16, 64, 7, 2, 30, 30), (2, 128, 128, 3, 1, 12, 12),
                                                 (2, 64, 64, 7, 2, 96, 96), (2, 64, 64, 3, 2, 80, 96)])
def test_training_convolution_function_vs_torch(B, cin, cout, k, s, H, W, monkeypatch, prec):
    """(prec: LAV_TRAIN_PRECISION - round 5's three bf16 pieces, or round 6's two fp16 pieces with the activations' maxima measured
    once per tensor and handed to forward / data gradient / weight gradient.)
    lav_amd.train.hipnn.conv2d - forward on lav_conv2d over the live parameter (device-side repack), data gradient on the adjoint
    lav_conv2d plan (the transposed convolution with the same weights, strides 1 and 2), weight gradient on lav_conv_wgrad where it applies - against torch's own
    convolution and its autograd, values and all gradients within 1e-4 of the largest reference value; a second call after the weight
    changed in place must see the new weights (the packed buffer is re-gathered on every forward)."""
    from lav_amd.train.hipnn import conv2d
    monkeypatch.setenv("LAV_TRAIN_PRECISION", prec)
    torch.manual_seed(B + cin + k)
    x = torch.randn((B, cin, H, W), device=DEV, requires_grad=True)
    w = (torch.randn((cout, cin, k, k), device=DEV) / (cin * k * k) ** 0.5).requires_grad_(True)
    for rep in range(2):
        y = conv2d(x, w, s, (k // 2, k // 2))
        dy = torch.randn_like(y)
        gx, gw = torch.autograd.grad(y, (x, w), dy)
        xr, wr = x.detach().cpu().double().requires_grad_(True), w.detach().cpu().double().requires_grad_(True)
        yr = torch.nn.functional.conv2d(xr, wr, None, s, k // 2)
        gxr, gwr = torch.autograd.grad(yr, (xr, wr), dy.cpu().double())
        for name, got, want in (("y", y, yr), ("dx", gx, gxr), ("dw", gw, gwr)):
            tol = 1e-4 * want.abs().max().item()
            assert (got.detach().cpu().double() - want.detach()).abs().max().item() < tol, (name, rep)
        with torch.no_grad():
            w.mul_(0.5).add_(0.01)


@pytest.mark.parametrize("B,cin,cout,k,s,p,op,H,W", [(2, 128, 128, 4, 2, 1, 0, 20, 24), (2, 128, 64, 4, 4, 0, 0, 10, 12), (3, 64, 128, 3, 2, 1, 1, 16, 16)])
def test_training_transposed_convolution_function_vs_torch(B, cin, cout, k, s, p, op, H, W, monkeypatch):
    """lav_amd.train.hipnn.conv_transpose_module with LAV_TRAIN_CONVT=hip (opt-in: 1 ms per step slower than MIOpen on the backbone's
    up-convolutions): forward on lav_conv2d's transposed plan over
    the live parameter, data gradient = the ordinary convolution with the same tensor, weight gradient torch - against torch's own
    ConvTranspose2d and its autograd in float64 (1e-4 of the largest reference value); a changed weight is seen by the next call."""
    from lav_amd.train.hipnn import _ConvT2d, conv_transpose_module
    monkeypatch.setenv("LAV_TRAIN_CONVT", "hip")
    torch.manual_seed(B + cin + k)
    m = torch.nn.ConvTranspose2d(cin, cout, k, s, p, op, bias=False).to(DEV)
    x = torch.randn((B, cin, H, W), device=DEV, requires_grad=True)
    for rep in range(2):
        y = conv_transpose_module(m, x)
        assert y.grad_fn is not None and type(y.grad_fn).__name__.startswith(_ConvT2d.__name__), "the liblav_amd function must be the one that ran"
        dy = torch.randn_like(y)
        gx, gw = torch.autograd.grad(y, (x, m.weight), dy)
        xr, wr = x.detach().cpu().double().requires_grad_(True), m.weight.detach().cpu().double().requires_grad_(True)
        yr = torch.nn.functional.conv_transpose2d(xr, wr, None, s, p, op)
        gxr, gwr = torch.autograd.grad(yr, (xr, wr), dy.cpu().double())
        for name, got, want in (("y", y, yr), ("dx", gx, gxr), ("dw", gw, gwr)):
            tol = 1e-4 * want.abs().max().item()
            assert (got.detach().cpu().double() - want.detach()).abs().max().item() < tol, (name, rep)
        with torch.no_grad():
            m.weight.mul_(0.5).add_(0.01)
