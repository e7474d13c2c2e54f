"""Shared test helpers: golden fixtures, seeded modules, comparison utilities."""
from __future__ import annotations

import os
import zlib

import numpy as np
import torch

from lav_amd import synth

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")
CFG = dict(min_x=-10, max_x=70, min_y=-40, max_y=40, pixels_per_meter=4)
Y_OFF = 1 + CFG["min_x"] / ((CFG["max_x"] - CFG["min_x"]) / 2)


def crc(a) -> int:
    return zlib.crc32(np.ascontiguousarray(a).tobytes())


class Golden:
    """Lazy access to tests/golden/*.npz (made by tests/golden/make_golden.py from the reference)."""

    def __init__(self):
        self._c = {}

    def __getitem__(self, name):
        if name not in self._c:
            self._c[name] = dict(np.load(os.path.join(GOLD, name + ".npz")))
        return self._c[name]


def build_models(device="cpu"):
    """Our LiDARModel / UniPlanner with the same seeded weights make_golden.py gave the reference."""
    import lav_amd
    lm = lav_amd.LiDARModel(num_input=16, backbone="cnn", num_features=[64, 64], **CFG)
    bp = lav_amd.BEVPlanner(pixels_per_meter=4, crop_size=96, feature_x_jitter=1.5, feature_angle_jitter=20,
                            x_offset=0, y_offset=Y_OFF, num_cmds=6, num_plan=20, num_plan_iter=5, num_frame_stack=2)
    up = lav_amd.UniPlanner(bp, pixels_per_meter=4, crop_size=96, feature_x_jitter=1.5, feature_angle_jitter=20,
                            x_offset=0, y_offset=Y_OFF, num_cmds=6, num_plan=20, num_input_feature=384, num_plan_iter=5)
    lm.load_state_dict(synth.seeded_state_dict(lm, prefix="lidar."))
    up.load_state_dict(synth.seeded_state_dict(up, prefix="uni."))
    return lm.eval().to(device), up.eval().to(device)


_sd_cache = {}


def state_dicts():
    """(lidar_sd, uni_sd) as CPU float tensors, keys as in the reference checkpoints."""
    if not _sd_cache:
        lm, up = build_models("cpu")
        _sd_cache["l"] = {k: v.clone() for k, v in lm.state_dict().items()}
        _sd_cache["u"] = {k: v.clone() for k, v in up.state_dict().items()}
    return _sd_cache["l"], _sd_cache["u"]


def pointnet_sd_numpy():
    lsd, _ = state_dicts()
    return {k[len("point_pillar_net."):]: v.numpy() for k, v in lsd.items() if k.startswith("point_pillar_net.")}


def sub_sd(sd, prefix):
    return {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}


def assert_close(a, b, atol, rtol=0.0, what=""):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    assert a.shape == b.shape, f"{what}: shape {a.shape} vs {b.shape}"
    err = np.abs(a - b)
    tol = atol + rtol * np.abs(b)
    bad = err > tol
    assert not bad.any(), f"{what}: {bad.sum()} / {bad.size} beyond tol; max err {err.max():.3e} (|ref| max {np.abs(b).max():.3e})"


# ---------------------------------------------------------------------------------------------- data-loader fixture
DATASET_CASES = (("temporal_bev", (0, 9, 13)), ("bev", (3,)), ("lidar", (10,)), ("lidar_painted", (17,)), ("temporal_lidar_painted", (0, 3, 12)))


def dataset_fixture_config(root, routes=2, frames=30):
    """Route data + YAML of the loader fixture (tests/golden/make_golden.py:gold_datasets and tests/test_data_host.py build the
    same thing): synthetic routes written by lav_amd.data.synthetic_route (seeded), the data-loader keys of the reference's
    config_v2.yaml (tests/golden/dataset_config.yaml) with data_dir pointed at them."""
    import yaml
    from lav_amd.data import synthetic_route
    synthetic_route.make_dataset(os.path.join(root, "data"), routes=routes, frames=frames, seed=0, points=2500)
    with open(os.path.join(GOLD, "dataset_config.yaml")) as f:
        cfg = yaml.safe_load(f)
    cfg["data_dir"] = os.path.join(root, "data")
    path = os.path.join(root, "config.yaml")
    with open(path, "w") as f:
        yaml.safe_dump(cfg, f)
    return path
