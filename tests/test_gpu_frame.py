"""Parity of the per-frame glue kernels (lav_merge_ticks, lav_stack_sweeps, lav_extract_peaks; through the C ABI)
against the oracle's restatement of lav_agent_fast.py / model_inference.py."""
import math

import numpy as np
import pytest
import torch

from lav_amd import ops, synth
from oracle import bev as obev
from oracle import frame as oframe

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda")


def test_merge_ticks_concat_egobox_prev():
    P = 5000
    tick = synth.lidar_sweep(P, name="mt0")
    prev = synth.lidar_sweep(P, name="mt1")
    tick[:40, :3] = np.array([-1.0, 0.1, -1.2], np.float32)      # inside the ego box
    prev[7, :3] = np.array([-2.4, 0.0, -1.2], np.float32)        # on the box face: kept (strict inequalities)
    prev[8, :3] = np.array([-2.39, 0.79, -1.01], np.float32)     # inside
    t, p = torch.from_numpy(tick).to(DEV), torch.from_numpy(prev).to(DEV)
    cur = ops.merge_ticks(t, p).cpu().numpy()
    ref = np.concatenate([tick, prev])
    x, y, z = ref[:, 0], ref[:, 1], ref[:, 2]
    ego = (x > -2.4) & (x < 0) & (y > -0.8) & (y < 0.8) & (z > -1.5) & (z < -1)      # lav_agent_fast.py:450-452
    assert ego.sum() >= 41 and not ego[P + 7] and ego[P + 8]
    np.testing.assert_array_equal(np.isnan(cur[:, 0]), ego)
    np.testing.assert_array_equal(cur[~ego], ref[~ego])
    np.testing.assert_array_equal(cur[ego, 1:], ref[ego, 1:])
    np.testing.assert_array_equal(p.cpu().numpy(), tick)         # prev_lidar = lidar


def test_stack_sweeps_matches_oracle():
    rows, slots = 3000, 15
    rng = np.random.default_rng(5)
    ring = rng.standard_normal((slots, rows, 8)).astype(np.float32) * 20
    fused = rng.standard_normal((rows, 8)).astype(np.float32) * 20
    fused[5, 0] = np.nan
    locs = {t: np.array([0.31 * t, -0.07 * t]) for t in range(slots)}
    oris = {t: 0.021 * t for t in range(slots)}
    slot, sweeps = 12, [12, 7, 2]
    loc0, ori0 = locs[12], oris[12]
    Rs, ts = [], []
    for t in sweeps:
        dloc = (locs[t] - loc0) @ np.array([[math.cos(ori0), -math.sin(ori0)], [math.sin(ori0), math.cos(ori0)]])
        o = oris[t] - ori0
        Rs.append([[math.cos(o), math.sin(o), 0.], [-math.sin(o), math.cos(o), 0.], [0., 0., 1.]])
        ts.append([dloc[0], dloc[1], 0.])
    d_ring = torch.from_numpy(ring).to(DEV)
    out = ops.stack_sweeps(torch.from_numpy(fused).to(DEV), d_ring, torch.tensor([slot], device=DEV),
                           torch.tensor(sweeps, device=DEV), torch.tensor(Rs, dtype=torch.float32, device=DEV),
                           torch.tensor(ts, dtype=torch.float32, device=DEV)).cpu().numpy()
    hist = {t: ring[t] for t in range(slots)}
    hist[12] = fused
    want = oframe.stack([hist[t] for t in range(13)], [locs[t] for t in range(13)], [oris[t] for t in range(13)])
    assert out.shape == want.shape == (3 * rows, 11)
    # xyz: the oracle multiplies with a BLAS matmul (summation order / FMA unspecified): 1-ulp class differences only
    np.testing.assert_allclose(out[:, :3], want[:, :3], rtol=0, atol=2e-5, equal_nan=True)
    np.testing.assert_array_equal(out[:, 3:], want[:, 3:])
    np.testing.assert_array_equal(d_ring[slot].cpu().numpy(), fused)                  # history write
    keep = [t for t in range(slots) if t != slot]
    np.testing.assert_array_equal(d_ring[keep].cpu().numpy(), ring[keep])


@pytest.mark.parametrize("shape,seed", [((2, 320, 320), 0), ((2, 320, 320), 1), ((1, 37, 53), 2), ((3, 8, 300), 3)])
def test_extract_peaks_matches_oracle(shape, seed):
    g = torch.Generator().manual_seed(seed)
    ncls, H, W = shape
    logits = torch.randn(shape, generator=g) * 2
    # smooth blobs so that NMS has real structure, plus flat plateaus (ties: every plateau pixel survives the NMS)
    logits = torch.nn.functional.avg_pool2d(logits[None], 5, 1, 2)[0] * 4
    logits[0, 2:5, 2:6] = 9.0
    size = torch.randn((2, H, W), generator=g)
    ori = torch.randn((2, H, W), generator=g)
    rows = ops.extract_peaks(logits.to(DEV), size.to(DEV), ori.to(DEV), apply_sigmoid=True).cpu()
    rows2 = ops.extract_peaks(torch.sigmoid(logits).to(DEV), size.to(DEV), ori.to(DEV)).cpu()
    hm = torch.sigmoid(logits)
    for c in range(ncls):
        score, loc = obev.extract_peak(hm[c])
        for r in (rows, rows2):
            got_loc = (r[c, :, 2] * W + r[c, :, 1]).long()
            # ties between equal scores may be ordered differently by topk: compare as sets within equal-score groups
            np.testing.assert_allclose(r[c, :, 0].numpy(), score.numpy(), rtol=0, atol=2e-7)
            for s in torch.unique(score):
                m = score == s
                assert sorted(got_loc[m].tolist()) == sorted(loc[m].tolist()) or m.sum() > 1
            ys, xs = got_loc // W, got_loc % W
            np.testing.assert_array_equal(r[c, :, 3:5].numpy(), size[:, ys, xs].T.numpy())
            np.testing.assert_array_equal(r[c, :, 5:7].numpy(), ori[:, ys, xs].T.numpy())
    # run-to-run determinism and counter reset: a second launch on the same workspace gives identical rows
    again = ops.extract_peaks(logits.to(DEV), size.to(DEV), ori.to(DEV), apply_sigmoid=True).cpu()
    np.testing.assert_array_equal(again.numpy(), rows.numpy())


def test_extract_peaks_few_candidates():
    """A monotone ramp has a handful of NMS survivors: the remaining rows carry the suppressed score."""
    H = W = 16
    hm = (torch.arange(H * W, dtype=torch.float32).view(1, H, W) / (H * W))
    z = torch.zeros((2, H, W))
    rows = ops.extract_peaks(hm.to(DEV), z.to(DEV), z.to(DEV), max_det=15).cpu()
    score, loc = obev.extract_peak(hm[0])
    n = int((score > -1e4).sum())
    assert 1 <= n < 15
    np.testing.assert_array_equal(rows[0, :n, 0].numpy(), score[:n].numpy())
    assert (rows[0, :n, 2] * W + rows[0, :n, 1]).long().tolist() == loc[:n].tolist()
    assert (rows[0, n:, 0] < -1e4).all()


def test_det_decode_on_device_matches_host_rules():
    """lav_det_decode (who the other vehicles are, where, how many - decided in HBM) against the host decode that the
    reference goldens pin (InferModel.det_decode_fast: model_inference.py:95-144), on random peak rows that exercise every
    filter: score threshold, the two range limits around the hard-coded ego pixel, the box-size rule, the ego's own box."""
    import lav_amd
    from tests.util import build_models
    lm, up = build_models(DEV)
    im = lav_amd.InferModel(lm, up, 1.5, 2.4, device=DEV)
    rng = np.random.default_rng(3)
    ox, oy = up.offsets()
    H, W = im._bev_hw
    centre = (float(W / 2 + ox * W / 2), float(H / 2 + oy * H / 2))
    actors = torch.zeros(45, device=DEV)
    n_out = torch.zeros(1, dtype=torch.int32, device=DEV)
    seen = set()
    for trial in range(40):
        rows = np.zeros((2, 15, 7), np.float32)
        rows[..., 0] = np.sort(rng.uniform(0.0, 1.0, (2, 15)).astype(np.float32) ** (1 + trial % 3), axis=1)[:, ::-1]
        rows[..., 1] = rng.integers(0, 320, (2, 15)); rows[..., 2] = rng.integers(0, 320, (2, 15))
        rows[1, :4, 1] = 160 + rng.integers(-5, 6, 4); rows[1, :4, 2] = 280 + rng.integers(-5, 6, 4)   # around the ego pixel
        rows[..., 3:5] = rng.uniform(0.0, 3.0, (2, 15, 2)); rows[1, 5, 3:5] = 0.39                        # below the 0.4 px box limit
        rows[..., 5:7] = rng.normal(0, 1, (2, 15, 2))
        ops.det_decode(torch.from_numpy(rows).to(DEV), actors, n_out, cls=1, min_score=0.2, ego_xy=(160, 280), near_px=2.0,
                       far_px=30 * im.pixels_per_meter, min_box=0.1 * im.pixels_per_meter, centre_xy=centre, skip_px=4.0,
                       ppm=up.pixels_per_meter)
        _, locs, oris = im.det_decode_fast(rows)
        n = int(n_out.cpu()[0])
        assert n == len(locs)
        a = actors.cpu().numpy()
        np.testing.assert_array_equal(a[:2 * n].reshape(-1, 2), locs.astype(np.float32))
        np.testing.assert_allclose(a[30:30 + n], oris.astype(np.float32), rtol=0, atol=2.4e-7)   # double atan2, one float32 ulp
        assert not a[2 * n:30].any() and not a[30 + n:].any()
        seen.add(n)
    assert len(seen) >= 5


def test_det_decode_reports_rows_and_count_to_pinned_host_memory():
    """lav_det_decode_report: the same launch leaves the peak rows and the count in pinned host memory and then bumps a sequence word
    (what the frame polls instead of two device->host copies and an event): once the word has moved, rows and count are the
    device's, bit for bit, launch after launch - eagerly and replayed from a HIP graph."""
    import time
    rng = np.random.default_rng(5)
    actors = torch.zeros(45, device=DEV)
    n_out = torch.zeros(1, dtype=torch.int32, device=DEV)
    h_rows = torch.zeros((2, 15, 7), dtype=torch.float32).pin_memory()
    h_n = torch.full((1,), -1, dtype=torch.int32).pin_memory()
    h_seq = torch.zeros((1,), dtype=torch.int32).pin_memory()
    d_rows = torch.zeros((2, 15, 7), device=DEV)
    kw = dict(cls=1, min_score=0.2, ego_xy=(160, 280), near_px=2.0, far_px=120.0, min_box=0.4, centre_xy=(160.0, 240.0), skip_px=4.0, ppm=4.0)

    def wait(seq0):
        t0 = time.perf_counter()
        while int(h_seq.numpy()[0]) == seq0:
            assert time.perf_counter() - t0 < 10.0, "the sequence word never moved"

    def fresh():
        rows = rng.uniform(0.0, 1.0, (2, 15, 7)).astype(np.float32)
        rows[..., 1:3] = rng.integers(0, 320, (2, 15, 2))
        return rows
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        ops.det_decode(d_rows, actors, n_out, report=(h_rows, h_n, h_seq), **kw)
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        ops.det_decode(d_rows, actors, n_out, report=(h_rows, h_n, h_seq), **kw)
    torch.cuda.synchronize()
    for trial in range(30):
        rows = fresh()
        d_rows.copy_(torch.from_numpy(rows))
        torch.cuda.synchronize()
        seq0 = int(h_seq.numpy()[0])
        if trial % 2:
            g.replay()
        else:
            ops.det_decode(d_rows, actors, n_out, report=(h_rows, h_n, h_seq), **kw)
        wait(seq0)                                            # no synchronize: the word is the only hand-off
        got_rows, got_n = h_rows.numpy().copy(), int(h_n.numpy()[0])
        assert int(h_seq.numpy()[0]) == seq0 + 1
        np.testing.assert_array_equal(got_rows, rows)
        assert got_n == int(n_out.cpu()[0])
    # without the report the plain entry point leaves the host words alone
    seq0 = int(h_seq.numpy()[0])
    ops.det_decode(d_rows, actors, n_out, **kw)
    torch.cuda.synchronize()
    assert int(h_seq.numpy()[0]) == seq0
    with pytest.raises(RuntimeError):
        ops.det_decode(d_rows, actors, n_out, report=(torch.zeros(2, 15, 7), h_n, h_seq), **kw)   # not pinned


def test_batch_limit_skips_rows_on_the_device():
    """ops.batch_limit: conv / crop / cast launches of a capacity-sized batch leave the rows beyond the device-resident count
    untouched and compute the live rows exactly as an unlimited launch does."""
    import lav_amd
    from tests.util import build_models
    lm, up = build_models(DEV)
    g = torch.Generator().manual_seed(9)
    feats = torch.randn((1, 384, 160, 160), generator=g).to(DEV)
    locs = (torch.rand((15, 2), generator=g) * 20 - 10).to(DEV)
    oris = (torch.rand((15,), generator=g) * 6 - 3).to(DEV)

    def branch():
        crops = up.crop_feature(feats.expand(15, -1, -1, -1), locs, oris, up.pixels_per_meter / 2, up.crop_size)
        embd = up.lidar_conv_emb(crops)
        return crops, embd, up.cast(embd, mode="other")
    full = [t.clone() for t in branch()]
    for n in (0, 1, 7, 15):
        d_n = torch.tensor([n], dtype=torch.int32, device=DEV)
        marks = []
        with ops.batch_limit(d_n):
            crops, embd, cast = branch()
        for got, want in zip((crops, embd, cast), full):
            assert torch.equal(got[:n], want[:n]), f"live rows differ at n={n}"
    # a second unlimited run is unaffected by the limit having been used
    again = branch()
    assert all(torch.equal(a, b) for a, b in zip(again, full))


def test_module_level_extract_peak_keeps_the_reference_signature():
    """model_inference.extract_peak(heatmap, max_pool_ks, min_score, max_det, break_tie) -> [(score, x, y)], as
    team_code_v2/model_inference.py:189-202, on the HIP peak kernel."""
    from lav_amd.model_inference import extract_peak
    g = torch.Generator().manual_seed(2)
    hm = torch.sigmoid(torch.nn.functional.avg_pool2d(torch.randn((1, 1, 64, 48), generator=g) * 3, 5, 1, 2)[0, 0] * 4)
    got = extract_peak(hm.to(DEV), min_score=0.3)
    score, loc = obev.extract_peak(hm)
    want = [(float(s), int(l) % 48, int(l) // 48) for s, l in zip(score, loc) if s > 0.3]
    assert len(got) == len(want) > 0
    for (s, x, y), (ws, wx, wy) in zip(got, want):
        assert abs(s - ws) < 2e-7 and (x, y) == (wx, wy)


def test_frame_knobs_do_not_change_the_frame(tmp_path):
    """The launch-plumbing knobs of round 6 choose HOW a frame is enqueued, never what it computes: the default (peak rows reported through
    pinned memory, pose block in the staging launch's arguments, the brake net as two graphs) against every knob switched back
    (LAV_DET_REPORT=0 LAV_POSE_BLOCK=0 LAV_BRAKE_SPLIT=0: copies + event, an upload of its own, one brake graph), each in a process of
    its own (the knobs are read at import): the brake prediction, the plan, the detections' rows and the other vehicles' forecasts of
    four frames agree bit for bit."""
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    worker = tmp_path / "frames.py"
    worker.write_text(
        "import sys, numpy as np, torch\n"
        f"sys.path.insert(0, {repo!r})\n"
        "import bench\n"
        "dev = torch.device('cuda', 0)\n"
        "pipe, _, _ = bench.build_pipeline(dev)\n"
        "host, d = bench.synthetic_inputs(dev)\n"
        "out = {}\n"
        "for i in range(5):\n"
        "    loc, ori = bench.pose(i)\n"
        "    o = pipe.step(d['ticks'][i % len(d['ticks'])], d['all_rgbs'], d['rgbs'], d['tel_rgbs'], loc, ori, d['nxp'], 3)\n"
        "    torch.cuda.synchronize()\n"
        "    if o is None: continue\n"
        "    out[f'bra{i}'] = o['pred_bra'].cpu().numpy(); out[f'plan{i}'] = o['ego_plan_locs'].cpu().numpy()\n"
        "    out[f'cast{i}'] = o['other_cast_locs'].cpu().numpy(); out[f'det{i}'] = pipe.hn_det.copy()\n"
        "np.savez(sys.argv[1], **out)\n")
    got = {}
    for name, env in (("default", {}), ("knobs_off", {"LAV_DET_REPORT": "0", "LAV_POSE_BLOCK": "0", "LAV_BRAKE_SPLIT": "0"})):
        f = tmp_path / f"{name}.npz"
        r = subprocess.run([sys.executable, str(worker), str(f)], env={**os.environ, **env}, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        got[name] = dict(np.load(f))
    assert got["default"].keys() == got["knobs_off"].keys() and len(got["default"]) == 16
    for k in got["default"]:
        np.testing.assert_array_equal(got["default"][k], got["knobs_off"][k], err_msg=k)
