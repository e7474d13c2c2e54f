"""Oracle: PointPillars dynamic voxelisation + PointNet + scatter (numpy, float32).

TEST INFRASTRUCTURE - see oracle/__init__.py.  Restates
/root/reference/lav/models/point_pillar.py (PointPillarNet / DynamicPointNet) and the
two torch_scatter ops it calls.  Integer outputs (cell coordinates, unique pillar
list, inverse map) are the bit-exact contract; float outputs carry the tolerance
written in tests/.
"""
from __future__ import annotations

import numpy as np

f32 = np.float32


def grid_locations(points: np.ndarray, min_x, max_x, min_y, max_y, ppm):
    """point_pillar.py:70-79.  keep = x in [min_x,max_x) & y in [min_y,max_y);
    cell = trunc((xy - min) * ppm) computed in float32 (sub, then mul - two
    roundings, no fused multiply-add).  Returns kept points, (n,2) int64 (xi, yi)
    and the indices of the kept points in the input."""
    x = points[:, 0].astype(f32)
    y = points[:, 1].astype(f32)
    keep = (x >= f32(min_x)) & (x < f32(max_x)) & (y >= f32(min_y)) & (y < f32(max_y))
    kept = np.nonzero(keep)[0]
    p = points[kept]
    cx = (p[:, 0].astype(f32) - f32(min_x)) * f32(ppm)
    cy = (p[:, 1].astype(f32) - f32(min_y)) * f32(ppm)
    coords = np.stack([cx, cy], axis=1).astype(np.int64)  # .long(): truncation toward zero
    return p, coords, kept


def pillar_generation(coords_bxy: np.ndarray):
    """point_pillar.py:81-85: coords.unique(return_inverse=True, dim=0).
    Rows (b, xi, yi) sorted lexicographically; inverse maps each point to its row."""
    if coords_bxy.shape[0] == 0:
        return np.zeros((0, 3), np.int64), np.zeros((0,), np.int64)
    uniq, inv = np.unique(coords_bxy, axis=0, return_inverse=True)
    return uniq.astype(np.int64), inv.reshape(-1).astype(np.int64)


def scatter_mean(src: np.ndarray, index: np.ndarray, n: int) -> np.ndarray:
    """torch_scatter.scatter_mean(src, index, dim=0) (call site point_pillar.py:62):
    float32 running sum in point order, divided by the per-row count."""
    out = np.zeros((n,) + src.shape[1:], f32)
    np.add.at(out, index, src.astype(f32))
    cnt = np.zeros((n,), f32)
    np.add.at(cnt, index, f32(1))
    return (out / np.maximum(cnt, f32(1))[:, None]).astype(f32)


def scatter_max(src: np.ndarray, index: np.ndarray, n: int) -> np.ndarray:
    """torch_scatter.scatter_max(src, index, dim=0)[0] (call site point_pillar.py:33)."""
    out = np.full((n,) + src.shape[1:], -np.inf, f32)
    np.maximum.at(out, index, src.astype(f32))
    out[~np.isfinite(out)] = 0
    return out


def decorate(points, uniq, inv, min_x, min_y, ppm):
    """point_pillar.py:55-68.  NB the reference pairs yi with min_x and xi with
    min_y (swapped) and uses un-centred cell origins; trained weights bake this in."""
    x_centers = uniq[inv][:, 2].astype(f32) / f32(ppm) + f32(min_x)
    y_centers = uniq[inv][:, 1].astype(f32) / f32(ppm) + f32(min_y)
    xyz = points[:, :3].astype(f32)
    cluster = xyz - scatter_mean(xyz, inv, uniq.shape[0])[inv]
    xp = xyz[:, 0] - x_centers
    yp = xyz[:, 1] - y_centers
    return np.concatenate([points.astype(f32), cluster, xp[:, None], yp[:, None]], axis=1)


def point_net(feats: np.ndarray, sd: dict, prefix: str = "point_net.net.", eps: float = 1e-5):
    """DynamicPointNet.net in eval mode (point_pillar.py:12-26): two blocks of
    Linear -> BatchNorm1d (running statistics) -> ReLU."""
    h = feats.astype(f32)
    for lin, bn in ((0, 1), (3, 4)):
        w = np.asarray(sd[f"{prefix}{lin}.weight"], f32)
        b = np.asarray(sd[f"{prefix}{lin}.bias"], f32)
        h = h @ w.T + b
        mu = np.asarray(sd[f"{prefix}{bn}.running_mean"], f32)
        var = np.asarray(sd[f"{prefix}{bn}.running_var"], f32)
        g = np.asarray(sd[f"{prefix}{bn}.weight"], f32)
        be = np.asarray(sd[f"{prefix}{bn}.bias"], f32)
        h = (h - mu) / np.sqrt(var + f32(eps)) * g + be
        h = np.maximum(h, f32(0))
    return h.astype(f32)


def scatter_points(feat: np.ndarray, uniq: np.ndarray, batch: int, nx: int, ny: int) -> np.ndarray:
    """point_pillar.py:87-90: canvas[b, :, clamp(ny-1-xi), clamp(yi)] = feat."""
    canvas = np.zeros((batch, feat.shape[1], ny, nx), f32)
    rows = np.clip(ny - 1 - uniq[:, 1], 0, ny - 1)
    cols = np.clip(uniq[:, 2], 0, nx - 1)
    canvas[uniq[:, 0], :, rows, cols] = feat
    return canvas


def pillar_forward(lidar_list, num_points, sd, min_x=-10, max_x=70, min_y=-40, max_y=40, ppm=4,
                   prefix: str = "point_net.net."):
    """PointPillarNet.forward (point_pillar.py:92-116).

    Returns dict(canvas (B,C,ny,nx), unique_coords (P,3), inverse (N_kept,),
    kept (N_kept,) index into the concatenated input, feat (P,C))."""
    nx = (max_x - min_x) * ppm
    ny = (max_y - min_y) * ppm
    coords, pts, kept_all = [], [], []
    base = 0
    for b, p in enumerate(lidar_list):
        p = np.asarray(p, f32)[: int(num_points[b])]
        kp, c, kept = grid_locations(p, min_x, max_x, min_y, max_y, ppm)
        coords.append(np.concatenate([np.full((len(c), 1), b, np.int64), c], axis=1))
        pts.append(kp)
        kept_all.append(kept + base)
        base += len(lidar_list[b])
    coords = np.concatenate(coords, axis=0)
    pts = np.concatenate(pts, axis=0)
    uniq, inv = pillar_generation(coords)
    dec = decorate(pts, uniq, inv, min_x, min_y, ppm)
    h = point_net(dec, sd, prefix)
    feat = scatter_max(h, inv, uniq.shape[0])
    canvas = scatter_points(feat, uniq, len(lidar_list), nx, ny)
    return dict(canvas=canvas, unique_coords=uniq, inverse=inv, kept=np.concatenate(kept_all),
                feat=feat, decorated=dec)
