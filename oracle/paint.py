"""Oracle: LiDAR -> camera projection and semantic point painting (numpy float32).

TEST INFRASTRUCTURE - see oracle/__init__.py.  Restates
/root/reference/team_code_v2/model_inference.py:44-50 (forward_paint), :75-93
(point_painting) and :255-297 (CoordConverter), plus the two CARLA matrices the
converter is built from (LibCarla geom/Transform.h; "parity unpinned", see
oracle/__init__.py).

Arithmetic contract shared with lav_amd/csrc/paint.hip: every matrix-vector
product is evaluated in float32, terms in k order, each multiply and each add
rounded separately (no fused multiply-add):  ((m0*x + m1*y) + m2*z) + m3*w.
The reference evaluates the same products with a BLAS sgemm whose association
is unspecified, so a projected coordinate that lands within an ulp of an
integer pixel boundary may truncate differently; tests count those points.
"""
from __future__ import annotations

import numpy as np

f32 = np.float32
CAMERA_YAWS = (-60.0, 0.0, 60.0)  # model_inference.py:12
INT_INVALID = np.iinfo(np.int64).min


def carla_matrix(x, y, z, yaw_deg=0.0):
    """carla.Transform(Location(x,y,z), Rotation(yaw=yaw)).get_matrix(), roll=pitch=0, float32."""
    a = f32(yaw_deg) * (f32(np.pi) / f32(180.0))
    c, s = np.cos(a, dtype=f32), np.sin(a, dtype=f32)
    return np.array([[c, -s, 0, x], [s, c, 0, y], [0, 0, 1, z], [0, 0, 0, 1]], f32)


def carla_inverse_matrix(x, y, z, yaw_deg=0.0):
    """...get_inverse_matrix(): rotation transposed, translation = R^T (-t), float32 term by term."""
    a = f32(yaw_deg) * (f32(np.pi) / f32(180.0))
    c, s = np.cos(a, dtype=f32), np.sin(a, dtype=f32)
    ax, ay, az = -f32(x), -f32(y), -f32(z)
    one, zero = f32(1), f32(0)
    tx = ax * c + ay * s + az * zero
    ty = ax * (-s) + ay * c + az * (-zero)
    tz = ax * (-zero) + ay * zero + az * one
    return np.array([[c, s, 0, tx], [-s, c, 0, ty], [0, 0, 1, tz], [0, 0, 0, 1]], f32)


def camera_matrices(cam_yaw, lidar_xyz, cam_xyz, rgb_h=288, rgb_w=256, fov=64):
    """CoordConverter.__init__ (model_inference.py:255-278): K, lidar_to_world, world_to_cam as float32."""
    focal = rgb_w / (2.0 * np.tan(fov * np.pi / 360.0))
    K = np.eye(3, dtype=f32)
    K[0, 0] = K[1, 1] = f32(focal)
    K[0, 2] = f32(rgb_w / 2.0)
    K[1, 2] = f32(rgb_h / 2.0)
    l2w = carla_matrix(*lidar_xyz)
    w2c = carla_inverse_matrix(*cam_xyz, yaw_deg=cam_yaw)
    return K, l2w, w2c


def _mv4(m, x, y, z, w):
    return [((m[i, 0] * x + m[i, 1] * y) + m[i, 2] * z) + m[i, 3] * w for i in range(4)]


def _to_long(v):
    """Tensor.long() on float32: truncation toward zero; non-finite / out-of-range
    values become INT64_MIN as on x86-64 (they fail every validity test below)."""
    ok = np.isfinite(v) & (np.abs(v) < f32(2.0 ** 62))
    return np.where(ok, np.trunc(np.where(ok, v, 0)).astype(np.int64), INT_INVALID)


def project(lidar_xyz: np.ndarray, K, l2w, w2c):
    """CoordConverter.forward (model_inference.py:280-297) -> (n,3) int64 (u, v, depth)."""
    x, y, z = (lidar_xyz[:, i].astype(f32) for i in range(3))
    one = np.ones_like(x)
    wx, wy, wz, ww = _mv4(l2w, x, y, z, one)
    cx, cy, cz, _ = _mv4(w2c, wx, wy, wz, ww)
    X, Y, Z = cy, -cz, cx                                 # :289 axis swap
    p0 = (K[0, 0] * X + K[0, 1] * Y) + K[0, 2] * Z
    p1 = (K[1, 0] * X + K[1, 1] * Y) + K[1, 2] * Z
    p2 = (K[2, 0] * X + K[2, 1] * Y) + K[2, 2] * Z
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        den = f32(1e-5) + p2
        u = p0 / den
        v = p1 / den
    return np.stack([_to_long(u), _to_long(v), _to_long(p2)], axis=1), np.stack([u, v, p2], axis=1)


def point_painting(lidar: np.ndarray, sems: np.ndarray, cams):
    """InferModel.point_painting (model_inference.py:75-93).  sems (ncam, C, H, W);
    later cameras overwrite earlier ones; unpainted points stay 0."""
    n = len(lidar)
    ncam, c, h, w = sems.shape
    painted = np.zeros((n, c), f32)
    for sem, (K, l2w, w2c) in zip(sems, cams):
        uvz, _ = project(lidar[:, :3], K, l2w, w2c)
        u, v, z = uvz[:, 0], uvz[:, 1], uvz[:, 2]
        valid = (z >= 0) & (u >= 0) & (u < w) & (v >= 0) & (v < h)
        painted[valid] = sem[:, v[valid], u[valid]].T
    return painted


def forward_paint(cur_lidar: np.ndarray, pred_sem: np.ndarray, camera_x=1.5, camera_z=2.4):
    """InferModel.forward_paint (model_inference.py:44-50): (N,4)+(3,5,H,W) -> (N,8)."""
    cams = [camera_matrices(yaw, (0, 0, camera_z), (camera_x, 0, camera_z)) for yaw in CAMERA_YAWS]
    sem = pred_sem[:, 1:].astype(f32) * (f32(1) - pred_sem[:, :1].astype(f32))
    painted = point_painting(cur_lidar, sem, cams)
    return np.concatenate([cur_lidar.astype(f32), painted], axis=1)


def preprocess(lidar: np.ndarray):
    """LAVAgent.preprocess (team_code_v2/lav_agent_fast.py:450-457): drop the ego-vehicle box."""
    x, y, z = lidar[:, 0], lidar[:, 1], lidar[:, 2]
    idx = (x > f32(-2.4)) & (x < 0) & (y > f32(-0.8)) & (y < f32(0.8)) & (z > f32(-1.5)) & (z < -1)
    return lidar[~idx]
