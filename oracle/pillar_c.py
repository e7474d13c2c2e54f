"""ctypes wrapper of oracle/pillar_c.c (TEST INFRASTRUCTURE - see oracle/__init__.py)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "_build", "liboracle.so")


class Grid(C.Structure):
    _fields_ = [("min_x", C.c_float), ("max_x", C.c_float), ("min_y", C.c_float), ("max_y", C.c_float),
                ("ppm", C.c_float), ("nx", C.c_int), ("ny", C.c_int)]


class Net(C.Structure):
    _fields_ = [(n, C.c_void_p * 2) for n in ("w", "b", "bn_mean", "bn_var", "bn_gamma", "bn_beta")] + \
               [("num_input", C.c_int), ("channels", C.c_int), ("eps", C.c_float)]


def build(force: bool = False) -> str:
    src = os.path.join(HERE, "pillar_c.c")
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(src):
        subprocess.run(["make", "-C", HERE, "-B"], check=True, stdout=subprocess.DEVNULL)
    return LIB


def pillar_forward(points: np.ndarray, sd: dict, min_x=-10, max_x=70, min_y=-40, max_y=40, ppm=4,
                   prefix: str = "point_net.net.", eps: float = 1e-5):
    lib = C.CDLL(build())
    pts = np.ascontiguousarray(points, np.float32)
    n, D = pts.shape
    nx, ny = (max_x - min_x) * ppm, (max_y - min_y) * ppm
    g = Grid(min_x, max_x, min_y, max_y, ppm, nx, ny)
    keep = []
    net = Net()
    for l, (lin, bn) in enumerate(((0, 1), (3, 4))):
        for field, key in (("w", f"{lin}.weight"), ("b", f"{lin}.bias"), ("bn_mean", f"{bn}.running_mean"),
                           ("bn_var", f"{bn}.running_var"), ("bn_gamma", f"{bn}.weight"), ("bn_beta", f"{bn}.bias")):
            a = np.ascontiguousarray(sd[prefix + key], np.float32)
            keep.append(a)
            getattr(net, field)[l] = a.ctypes.data
    net.num_input, net.channels, net.eps = D + 5, keep[0].shape[0], eps
    Cc = net.channels
    canvas = np.empty((1, Cc, ny, nx), np.float32)
    uc = np.empty((max(n, 1), 3), np.int32)
    inv = np.empty((max(n, 1),), np.int32)
    cnt = np.zeros(2, np.int32)
    lib.oracle_pillar_forward(pts.ctypes.data_as(C.c_void_p), n, D, C.byref(g), C.byref(net),
                              canvas.ctypes.data_as(C.c_void_p), uc.ctypes.data_as(C.c_void_p),
                              inv.ctypes.data_as(C.c_void_p), cnt.ctypes.data_as(C.c_void_p))
    return dict(canvas=canvas, unique_coords=uc[: cnt[0]].astype(np.int64), inverse=inv[: cnt[1]].astype(np.int64))
