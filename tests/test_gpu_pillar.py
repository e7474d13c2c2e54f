"""Parity of the HIP pillar pipeline (through the C ABI) against the oracle and the reference goldens.
BASELINE.json config #2: 32k-point sweeps -> 320x320x64 grid, bit-exact indices."""
import os

import numpy as np
import pytest
import torch

import lav_amd
from lav_amd import synth
from oracle import pillar as opillar
from tests.test_oracle_golden import GRID, pillar_cases
from tests.util import CFG, assert_close, pointnet_sd_numpy, state_dicts, sub_sd

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda")


@pytest.fixture(scope="module")
def ppn():
    lsd, _ = state_dicts()
    m = lav_amd.PointPillarNet(16, [64, 64], **CFG)
    m.load_state_dict(sub_sd(lsd, "point_pillar_net."))
    return m.eval().to(DEV)


def run(ppn, clouds, n, impl="mfma"):
    os.environ["LAV_PILLAR_IMPL"] = impl
    try:
        lst = [torch.from_numpy(c).to(DEV) for c in clouds]
        canvas, uc, inv = ppn(lst, n, return_indices=True)
        torch.cuda.synchronize()
    finally:
        os.environ.pop("LAV_PILLAR_IMPL", None)
    return canvas.cpu().numpy(), uc.cpu().numpy(), inv.cpu().numpy()


@pytest.mark.parametrize("impl", ["mfma", "valu"])
@pytest.mark.parametrize("name", ["lidar", "uniform", "edge", "one_cell", "batch2"])
def test_pillar_vs_reference_golden(golden, ppn, name, impl):
    g = golden["pillar"]
    clouds, n = pillar_cases(g)[name]
    n = n or [len(c) for c in clouds]
    canvas, uc, inv = run(ppn, clouds, n, impl)
    np.testing.assert_array_equal(uc, g[f"{name}/unique_coords"])     # bit-exact vs the reference
    np.testing.assert_array_equal(inv, g[f"{name}/inverse"])
    ref = opillar.scatter_points(g[f"{name}/feat"], g[f"{name}/unique_coords"].astype(np.int64), len(clouds), 320, 320)
    assert_close(canvas, ref, atol=3e-5, rtol=2e-5, what=f"canvas[{name},{impl}]")


def test_pillar_empty_and_outside(ppn):
    for cloud in (np.zeros((0, 11), np.float32), np.full((64, 11), 100.0, np.float32)):
        canvas, uc, inv = run(ppn, [cloud], [len(cloud)])
        assert canvas.shape == (1, 64, 320, 320) and not canvas.any()
        assert len(uc) == 0 and len(inv) == 0


@pytest.mark.parametrize("n_per_sweep,kind", [(10923, "lidar"), (65536, "lidar"), (65536, "uniform")])
def test_pillar_full_size_vs_oracle(ppn, n_per_sweep, kind):
    """32 768-point (config #2) and agent-realistic 196 608-point clouds against the numpy oracle."""
    pts = synth.stacked_lidar(n_per_sweep, kind=kind)
    canvas, uc, inv = run(ppn, [pts], [len(pts)])
    o = opillar.pillar_forward([pts], [len(pts)], pointnet_sd_numpy(), *GRID)
    np.testing.assert_array_equal(uc, o["unique_coords"])
    np.testing.assert_array_equal(inv, o["inverse"])
    assert_close(canvas, o["canvas"], atol=5e-5, rtol=2e-5, what="canvas")


def test_pillar_properties(ppn):
    """Size-independent properties: run-to-run determinism, invariance to point order (the per-cell sums are
    exact fixed point and the max is order free, so the canvas must be BIT-identical), idempotent workspace."""
    pts = synth.stacked_lidar(65536, kind="lidar")
    a, uca, _ = run(ppn, [pts], [len(pts)])
    b, _, _ = run(ppn, [pts], [len(pts)])
    assert np.array_equal(a, b)
    perm = np.random.Generator(np.random.PCG64(1)).permutation(len(pts))
    c, ucc, _ = run(ppn, [pts[perm]], [len(pts)])
    assert np.array_equal(a, c)
    np.testing.assert_array_equal(uca, ucc)
    # occupied cells <-> non-zero canvas columns
    occ = np.zeros((320, 320), bool)
    occ[np.clip(319 - uca[:, 1], 0, 319), np.clip(uca[:, 2], 0, 319)] = True
    assert not a[0][:, ~occ].any()
    # num_points truncation == physically truncated cloud
    d, _, _ = run(ppn, [pts], [50000])
    e, _, _ = run(ppn, [pts[:50000]], [50000])
    assert np.array_equal(d, e)
    # the pairing hint of k_rows (per-class loads of the PREVIOUS call) moves time, never results: a stale hint from a very
    # different cloud and the cloud's own hint give the same bits
    other = pts.copy()
    other[:, 0], other[:, 1] = 60.0 - pts[:, 0], -pts[:, 1]
    run(ppn, [pts], [len(pts)])
    f, ucf, _ = run(ppn, [other], [len(other)])
    g, ucg, _ = run(ppn, [other], [len(other)])
    assert np.array_equal(f, g)
    np.testing.assert_array_equal(ucf, ucg)


def test_pillar_336_grid():
    """BASELINE.json words the grid as 336x336x64 (84 m range); parity is defined by the same oracle."""
    lsd, _ = state_dicts()
    cfg = dict(min_x=-10, max_x=74, min_y=-42, max_y=42, pixels_per_meter=4)
    m = lav_amd.PointPillarNet(16, [64, 64], **cfg)
    m.load_state_dict(sub_sd(lsd, "point_pillar_net."))
    m = m.eval().to(DEV)
    pts = synth.stacked_lidar(10923, kind="uniform")
    canvas, uc, inv = m([torch.from_numpy(pts).to(DEV)], [len(pts)], return_indices=True)
    o = opillar.pillar_forward([pts], [len(pts)], pointnet_sd_numpy(), -10, 74, -42, 42, 4)
    assert canvas.shape == (1, 64, 336, 336)
    np.testing.assert_array_equal(uc.cpu().numpy(), o["unique_coords"])
    np.testing.assert_array_equal(inv.cpu().numpy(), o["inverse"])
    assert_close(canvas.cpu().numpy(), o["canvas"], atol=5e-5, rtol=2e-5, what="canvas336")


@pytest.mark.parametrize("impl", ["mfma", "valu"])
def test_pillar_canvas_bound_for_the_first_bev_layer(golden, ppn, impl):
    """lav_pillar_scatter_amax (round 6): the per-workgroup maxima the canvas kernel leaves for LAV_CONV_F16X3 readers.  They bound
    the canvas (on clouds without clamp-layer collisions: exactly its maximum), an empty canvas leaves zeros, and asking for them
    changes no bit of the canvas."""
    from lav_amd import ops
    os.environ["LAV_PILLAR_IMPL"] = impl
    try:
        for name in ("lidar", "uniform", "edge", "one_cell", "batch2"):
            clouds, n = pillar_cases(golden["pillar"])[name]
            n = n or [len(c) for c in clouds]
            lst = [torch.from_numpy(c).to(DEV) for c in clouds]
            with_b = ppn(lst, n)
            am = ops.amax_of(with_b)
            assert am is not None and am.count >= 1
            parts = am.buf[:am.count].cpu().numpy()
            os.environ["LAV_PILLAR_AMAX"] = "0"
            try:
                plain = ppn(lst, n)
            finally:
                os.environ.pop("LAV_PILLAR_AMAX")
            assert ops.amax_of(plain) is None
            assert torch.equal(with_b, plain)
            top = float(with_b.max())
            assert np.isfinite(parts).all() and (parts >= 0).all()
            assert parts.max() >= top, f"{name}: the parts' maximum {parts.max()} is below the canvas' {top}"
            if name in ("lidar", "uniform", "batch2"):     # (edge / one_cell exercise the clamp layers, where a replaced pillar may count)
                assert parts.max() == top
        empty = ppn([torch.full((64, 11), 100.0, device=DEV)], [64])
        am = ops.amax_of(empty)
        assert not empty.any() and float(am.buf[:am.count].max()) == 0.0
    finally:
        os.environ.pop("LAV_PILLAR_IMPL", None)
