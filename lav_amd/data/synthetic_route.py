"""Writes a small synthetic route in the data collector's LMDB layout (the keys lav/utils/datasets reads: `len`, `town`,
and per frame t `id_`, `loc_`, `ori_`, `bbox_`, `type_`, `nxp_`, `cmd_`, `bra_`, `lidar_`, `lidar_sem_`, `map_{0..11}_`).
No recorded data ships with the reference (Git-LFS pointers), so this is what the loader tests and a no-download smoke
run of the trainers read:

    python -m lav_amd.data.synthetic_route <data_dir> [--routes 2] [--frames 40]
    python train_bev_v2.py --config-path <yaml with data_dir: <data_dir>> ...
"""
from __future__ import annotations

import argparse
import os

import numpy as np

from . import image, lmdb_ro


def make_route(path: str, seed: int = 0, frames: int = 40, points: int = 2500, town: str = "Town01", sem_channels: int = 4) -> None:
    r = np.random.default_rng(seed)
    items = {b"len": str(frames).encode(), b"town": town.encode()}
    f32 = lambda a: np.asarray(a, np.float32).tobytes()
    # actors: ego (id 100, vehicle) drives a gentle arc; vehicles / pedestrians around it, some leave early
    n_veh, n_ped = 5, 3
    ids = np.array([100] + [200 + 7 * i for i in range(n_veh)] + [900 + 3 * i for i in range(n_ped)], np.int32)
    typ = np.array([1] + [1] * n_veh + [0] * n_ped, np.uint8)
    start = np.concatenate([[[10.0, -4.0]], r.uniform(-22, 22, (n_veh, 2)) + [10, -4], r.uniform(-8, 8, (n_ped, 2)) + [10, -4]])
    speed = np.concatenate([[0.35], r.uniform(0.0, 0.5, n_veh), r.uniform(0.0, 0.08, n_ped)])
    head = np.concatenate([[20.0], r.uniform(-180, 180, n_veh + n_ped)])
    turn = np.concatenate([[0.6], r.uniform(-1, 1, n_veh + n_ped)])
    box = np.concatenate([[[2.4, 1.0]], r.uniform([1.8, 0.8], [2.6, 1.1], (n_veh, 2)), r.uniform([0.3, 0.3], [0.5, 0.5], (n_ped, 2))])
    leaves_at = {int(ids[2]): frames // 2, int(ids[-1]): frames // 3}          # present in early frames only
    pos = start.copy()
    yy, xx = np.mgrid[0:320, 0:320]
    for t in range(frames):
        ang = head + turn * t
        keep = np.array([t < leaves_at.get(int(a), frames + 1) for a in ids])
        items[f"id_{t:05d}".encode()] = ids[keep].tobytes()
        items[f"loc_{t:05d}".encode()] = f32(pos[keep])
        items[f"ori_{t:05d}".encode()] = f32(ang[keep])
        items[f"bbox_{t:05d}".encode()] = f32(box[keep])
        items[f"type_{t:05d}".encode()] = typ[keep].tobytes()
        items[f"nxp_{t:05d}".encode()] = f32(pos[0] + 18 * np.array([np.cos(np.deg2rad(ang[0])), np.sin(np.deg2rad(ang[0]))]))
        items[f"cmd_{t:05d}".encode()] = np.array([(t // 7) % 6], np.uint8).tobytes()
        items[f"bra_{t:05d}".encode()] = np.array([int(t % 11 == 0)], np.uint8).tobytes()
        # LiDAR: ground ring + a few boxes, in the sensor frame; painted scores in [0, 1]
        rad = r.uniform(2.5, 45, points)
        az = r.uniform(-np.pi, np.pi, points)
        xyz = np.stack([rad * np.cos(az), rad * np.sin(az), r.normal(-2.2, 0.15, points), r.uniform(0, 1, points)], 1)
        xyz[: points // 20, :3] = r.uniform([-2.3, -0.7, -1.45], [-0.1, 0.7, -1.05], (points // 20, 3))          # the ego's own body
        items[f"lidar_{t:05d}".encode()] = f32(xyz)
        sem = r.uniform(0, 1, (points, sem_channels)) * (r.uniform(0, 1, (points, 1)) > 0.5)
        items[f"lidar_sem_{t:05d}".encode()] = f32(sem)
        # BEV maps (ego at pixel (160, 280), 4 px / m): 0 road, 1 vehicles, 2 pedestrians, 9 / 10 lane markings, others empty
        maps = np.zeros((12, 320, 320), np.uint8)
        maps[0][:, 110:210] = 255
        maps[0][230:270, :] = 255
        maps[9][:, 158:162] = 255
        maps[10][248:252, ::8] = 255
        c, s = np.cos(np.deg2rad(ang[0])), np.sin(np.deg2rad(ang[0]))
        for k in np.nonzero(keep)[0]:
            d = pos[k] - pos[0]
            fwd, lat = d[0] * c + d[1] * s, -d[0] * s + d[1] * c
            px, py = 160 + 4 * lat, 280 - 4 * fwd
            hw = 4 * box[k]
            m = (np.abs(xx - px) <= max(hw[1], 1)) & (np.abs(yy - py) <= max(hw[0], 1))
            maps[1 if typ[k] == 1 else 2][m] = 255
        for ch in range(12):
            items[f"map_{ch}_{t:05d}".encode()] = image.imencode_png(maps[ch])
        pos = pos + speed[:, None] * np.stack([np.cos(np.deg2rad(ang)), np.sin(np.deg2rad(ang))], 1)
    lmdb_ro.write(path, items.items())


def make_dataset(data_dir: str, routes: int = 2, frames: int = 40, seed: int = 0, points: int = 2500) -> None:
    os.makedirs(data_dir, exist_ok=True)
    towns = ["Town01", "Town03", "Town02", "Town06"]
    for i in range(routes):
        make_route(os.path.join(data_dir, f"route_{i:03d}"), seed=seed + i, frames=frames, points=points, town=towns[i % len(towns)])


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("data_dir")
    ap.add_argument("--routes", type=int, default=2)
    ap.add_argument("--frames", type=int, default=40)
    ap.add_argument("--points", type=int, default=2500)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    make_dataset(a.data_dir, a.routes, a.frames, a.seed, a.points)
    print(f"wrote {a.routes} route(s) of {a.frames} frames under {a.data_dir}")
