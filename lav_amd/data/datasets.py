"""Recorded routes -> training batches: the loaders behind `train_bev_v2.py` ('temporal_bev') and `train_full_v2.py`
('temporal_lidar_painted'), and their single-frame variants ('bev', 'lidar', 'lidar_painted').

Follows lav/utils/datasets/{basic,bev,temporal_bev,lidar,lidar_painted,temporal_lidar_painted}_dataset.py and
lav/utils/point_painting.py: same LMDB keys, same arithmetic (float32 reads, float64 geometry), same tuple layout, and the
same ORDER of random draws (torch.rand for the crop jitter / rotation, np.random for the stacked sweeps' pose jitter and
the point shuffle), so that a seeded reference loader and a seeded loader of this module return the same sample
(tests/test_data_host.py, against the reference's own classes run over stand-ins for `lmdb` and `cv2`).
Routes are read with lav_amd.data.lmdb_ro (no liblmdb here), images with lav_amd.data.image (no OpenCV here).

Differences from the reference, all deliberate: route directories are visited in sorted order (the reference takes
`glob` order, which is file-system dependent); the 'rgb' / 'seg' / 'bra' loaders (camera-model training, outside the two
trainers this repository mirrors) are not provided.
"""
from __future__ import annotations

import glob
import math
import os

import numpy as np
import torch
import yaml
from torch.utils.data import DataLoader, Dataset

from . import image, lmdb_ro

TRAIN_TOWNS = ("Town01", "Town03", "Town04", "Town06")
BEV_CENTER = (160, 280)          # ego pixel of the recorded 320x320 BEV maps: the centre of every rotation augment
MARGIN = 32                      # zero border added before a BEV map is shifted


# ---------------------------------------------------------------------------------------------------- route access
def read_array(txn, tag: str, t: int, dtype=np.float32, count: int = 1) -> np.ndarray:
    """`count` consecutive per-frame records `tag_{t:05d}` stacked (basic_dataset.py:80-82)."""
    return np.stack([np.frombuffer(txn.get(f"{tag}_{i:05d}".encode()), dtype) for i in range(t, t + count)])


def read_bev(txn, t: int, channels) -> np.ndarray:
    """(H, W, len(channels)) uint8: one grayscale PNG per map channel (basic_dataset.py:96-101)."""
    return np.stack([image.imdecode(np.frombuffer(txn.get(f"map_{c}_{t:05d}".encode()), np.uint8), image.IMREAD_GRAYSCALE) for c in channels], axis=-1)


def actor_tracks(txn, t0: int, T: int, max_pedestrian_radius: float, max_vehicle_radius: float):
    """The actors present in ALL frames t0 .. t0 + T and close enough at t0 (pedestrians type 0, vehicles type 1), as world
    frame tracks (basic_dataset.py:103-157).  Returns (ego_id, ego_locs, ego_oris, ego_bbox, present, locs, oris, bbox, typs),
    the last five dictionaries keyed by actor id."""
    ids0 = read_array(txn, "id", t0, np.int32).flatten()
    ego_id = ids0[0]
    present = {a: np.zeros(T + 1) for a in ids0}
    locs = {a: np.zeros((T + 1, 2)) for a in ids0}
    oris = {a: np.zeros(T + 1) for a in ids0}
    bbox = {a: np.zeros((T + 1, 2)) for a in ids0}
    typs = {a: np.zeros(T + 1) for a in ids0}
    for t in range(t0, t0 + T + 1):
        ids_t = read_array(txn, "id", t, np.int32).flatten()
        loc_t = read_array(txn, "loc", t).reshape(-1, 2)
        ori_t = read_array(txn, "ori", t).flatten()
        box_t = read_array(txn, "bbox", t).reshape(-1, 2)
        typ_t = read_array(txn, "type", t, np.uint8).flatten()
        for a, l, o, b, ty in zip(ids_t, loc_t, ori_t, box_t, typ_t):
            if a not in ids0:
                continue
            k = t - t0
            present[a][k] = 1
            locs[a][k] = l
            oris[a][k] = np.deg2rad(o)
            bbox[a][k] = b
            typs[a][k] = ty
    ego_locs, ego_oris, ego_bbox = locs[ego_id], oris[ego_id], bbox[ego_id]
    drop = {a for a, m in present.items() if not np.all(m)}
    for a in present:
        d = np.linalg.norm(locs[a][0] - ego_locs[0])
        if (typs[a][0] == 0 and d > max_pedestrian_radius) or (typs[a][0] == 1 and d > max_vehicle_radius):
            drop.add(a)
    for a in drop:
        for table in (present, typs, locs, oris, bbox):
            table.pop(a)
    return ego_id, ego_locs, ego_oris, ego_bbox, present, locs, oris, bbox, typs


def to_ego_frame(ego_locs, locs, oris, bbox, typs, ego_ori, T):
    """Tracks in the frame of the ego vehicle at t0, actors ordered by id (bev_dataset.py:97-114)."""
    origin = ego_locs[0]
    keys = sorted(list(locs.keys()))
    locs = np.array([locs[k] for k in keys]).reshape(-1, T, 2)
    oris = np.array([oris[k] for k in keys]).reshape(-1, T)
    bbox = np.array([bbox[k] for k in keys]).reshape(-1, T, 2)
    typs = np.array([typs[k] for k in keys]).reshape(-1, T)
    R = [[np.sin(ego_ori), np.cos(ego_ori)], [-np.cos(ego_ori), np.sin(ego_ori)]]
    return (ego_locs - origin) @ R, (locs - origin) @ R, oris - ego_ori, bbox, typs


# ---------------------------------------------------------------------------------------------------- augmentations
def rotate_image(img: np.ndarray, angle_deg: float, center=BEV_CENTER) -> np.ndarray:
    return image.warp_affine_linear(img, image.rotation_matrix_2d(center, angle_deg, 1.0))


def rotate_points(points, angle_deg: float, origin):
    r = np.deg2rad(angle_deg)
    return (points - origin) @ [[np.cos(r), np.sin(r)], [-np.sin(r), np.cos(r)]] + origin


def rotate_lidar(lidar, angle_deg: float):
    r = np.deg2rad(angle_deg)
    return lidar @ [[np.cos(r), np.sin(r), 0, 0], [-np.sin(r), np.cos(r), 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]]


def move_lidar_points(lidar, dloc, ori0, ori1):
    """A past sweep expressed in the current ego frame (temporal_lidar_painted_dataset.py:197-215)."""
    dloc = dloc @ [[np.cos(ori0), -np.sin(ori0)], [np.sin(ori0), np.cos(ori0)]]
    o = ori1 - ori0
    lidar = lidar @ [[np.cos(o), np.sin(o), 0, 0], [-np.sin(o), np.cos(o), 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]]
    lidar[:, :2] += dloc
    return lidar


class CameraProjection:
    """lidar point -> pixel of one RGB camera, in float64 with LibCarla's float32 transform entries
    (lav/utils/point_painting.py:5-47)."""

    def __init__(self, cam_yaw, lidar_xyz, cam_xyz, rgb_h, rgb_w, fov):
        from ..model_inference import carla_inverse_matrix, carla_matrix
        focal = rgb_w / (2.0 * np.tan(fov * np.pi / 360.0))
        self.K = np.identity(3)
        self.K[0, 0] = self.K[1, 1] = focal
        self.K[0, 2], self.K[1, 2] = rgb_w / 2.0, rgb_h / 2.0
        self.lidar_to_world = carla_matrix(*lidar_xyz).astype(np.float64)
        self.world_to_cam = carla_inverse_matrix(*cam_xyz, yaw_deg=cam_yaw).astype(np.float64)

    def pixels(self, lidar) -> np.ndarray:
        xyz = lidar[:, :3].T
        cam = self.world_to_cam @ (self.lidar_to_world @ np.r_[xyz, [np.ones(xyz.shape[1])]])
        uvz = self.K @ np.array([cam[1], -cam[2], cam[0]])
        return np.array([uvz[0] / (1e-5 + uvz[2]), uvz[1] / (1e-5 + uvz[2]), uvz[2]]).T.astype(int)


def paint_from_cameras(lidar, sems, cameras) -> np.ndarray:
    """Per point, the class scores of the pixel it projects to; a later camera overwrites an earlier one
    (point_painting.py:50-70)."""
    n_cls, h, w = sems[0].shape
    out = np.zeros((len(lidar), n_cls))
    for sem, cam in zip(sems, cameras):
        p = cam.pixels(lidar)
        u, v, z = p[:, 0], p[:, 1], p[:, 2]
        ok = (z >= 0) & (u >= 0) & (u < w) & (v >= 0) & (v < h)
        out[ok] = sem[:, v[ok], u[ok]].T
    return out


# ---------------------------------------------------------------------------------------------------- frame index
class RouteFrames(Dataset):
    """Every frame of every recorded route under `data_dir` that has num_plan future frames (basic_dataset.py:12-77).
    The YAML's keys become attributes, as in the reference."""

    def __init__(self, config_path, close_txn=False, seed=2021):
        super().__init__()
        with open(config_path, "r") as f:
            for key, value in yaml.safe_load(f).items():
                setattr(self, key, value)
        self.num_frames = 0
        self.txn_map, self.idx_map, self.dir_map = {}, {}, {}
        np.random.seed(seed)
        for route in sorted(glob.glob(f"{self.data_dir}/**")):
            if np.random.random() > self.percentage_data:        # the reference's per-route coin
                continue
            if not os.path.isfile(os.path.join(route, "data.mdb")):
                continue
            txn = lmdb_ro.open(route, max_readers=1, readonly=True, lock=False, readahead=False, meminit=False).begin(write=False)
            n = int(txn.get(b"len"))
            town = txn.get(b"town").decode()
            if not self.all_towns and town not in TRAIN_TOWNS:
                continue
            first = self.num_frames
            for i in range(n - self.num_plan):
                self.txn_map[first + i], self.idx_map[first + i], self.dir_map[first + i] = txn, i, route
            self.num_frames += max(n - self.num_plan, 0)
        self.nam_map = self.dir_map

    def __len__(self):
        return self.num_frames

    # shared pieces of the samples ---------------------------------------------------------------------------
    def _tracks(self, txn, t, vehicle_radius=None):
        return actor_tracks(txn, t, self.num_plan, self.max_pedestrian_radius,
                            self.max_vehicle_radius if vehicle_radius is None else vehicle_radius)

    def _pad_actors(self, locs, oris, typs):
        n = min(len(locs), self.max_objs)
        p_locs = np.zeros((self.max_objs, self.num_plan + 1, 2), np.float32)
        p_oris = np.zeros((self.max_objs,), np.float32)
        p_typs = np.zeros((self.max_objs,), np.int32)
        p_locs[:n], p_oris[:n], p_typs[:n] = locs[:n], oris[:n, 0], typs[:n, 0]
        return p_locs, p_oris, p_typs, n

    def _bev_stack(self, txn, index, angle, y_offset=0):
        """(3 + 2 (num_frame_stack + 1), 320, 320): road / lane channels of the current frame, then (vehicles, pedestrians) of
        the current and the stacked past frames moved into the current ego frame (temporal_bev_dataset.py:34-66)."""
        bev = np.zeros((3 + 2 * (self.num_frame_stack + 1), 320, 320), np.uint8)
        bev[:3] = self._bev_channels(txn, index, [0, 9, 10], angle_offset=angle, y_offset=y_offset)
        for k, i in enumerate(reversed(range(index - self.num_frame_stack, index + 1))):
            if i < 0:
                continue
            _, locs_i, oris_i, *_ = self._tracks(txn, i)
            if i == index:
                loc0, ori0 = locs_i[0], oris_i[0]
            dloc = (locs_i[0] - loc0) @ [[np.cos(ori0), -np.sin(ori0)], [np.sin(ori0), np.cos(ori0)]] * self.pixels_per_meter
            bev[3 + 2 * k:5 + 2 * k] = self._bev_channels(txn, i, [1, 2], angle=oris_i[0] - ori0, angle_offset=angle, y_offset=y_offset, loc=dloc)
        return bev

    def _bev_channels(self, txn, t, channels, angle=0, angle_offset=0, y_offset=0, loc=(0, 0)):
        dx, dy = map(int, loc)
        bev = rotate_image(read_bev(txn, t, channels), -angle * 180 / math.pi)
        bev = np.pad(bev, [[MARGIN, MARGIN], [MARGIN, MARGIN], [0, 0]])
        bev = bev[dx + MARGIN:dx + MARGIN + 320, dy + MARGIN + y_offset:dy + MARGIN + y_offset + 320, :]
        return (rotate_image(bev, angle_offset) > 0).astype(np.uint8).transpose(2, 0, 1)

    def _commands(self, txn, index):
        return (int(read_array(txn, "cmd", index, np.uint8).reshape(-1)[0]), int(read_array(txn, "bra", index, np.uint8).reshape(-1)[0]),
                read_array(txn, "nxp", index).reshape(2))


# ---------------------------------------------------------------------------------------------------- BEV loaders
class BEVDataset(RouteFrames):
    """'bev': (bev, ego_locs, cmd, nxp, bra, locs, oris, typs, num_objs) with a 5-channel map (bev_dataset.py:17-77)."""
    temporal = False

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.margin = MARGIN

    def __getitem__(self, idx):
        txn, index = self.txn_map[idx], self.idx_map[idx]
        _, ego_locs, ego_oris, _, _, locs, oris, bbox, typs = self._tracks(txn, index)
        ego_locs, locs, oris, bbox, typs = to_ego_frame(ego_locs, locs, oris, bbox, typs, ego_oris[0], self.num_plan + 1)
        offset = int((torch.rand(1) * 2 - 1) * self.x_jitter)
        offset = np.clip(offset, -MARGIN, MARGIN)
        angle = float(torch.rand(1) * 2 - 1) * self.angle_jitter
        if self.temporal:
            bev = self._bev_stack(txn, index, angle, y_offset=offset)
        else:
            bev = (rotate_image(read_bev(txn, index, [0, 1, 2, 9, 10]), angle) > 0).astype(np.uint8).transpose(2, 0, 1)
            bev = np.pad(bev, [[0, 0], [MARGIN, MARGIN], [MARGIN, MARGIN]])[:, MARGIN:MARGIN + 320, MARGIN + offset:MARGIN + offset + 320]
        shift = [offset / self.pixels_per_meter, 0]
        cmd, bra, nxp = self._commands(txn, index)
        if self.temporal:     # the two loaders rotate in a different order; the pivot is the (already moved) ego position
            locs = rotate_points(locs, -angle, ego_locs[0]) + shift
            oris[1:] = oris[1:] - np.deg2rad(angle)
            ego_locs = rotate_points(ego_locs, -angle, ego_locs[0]) + shift
            nxp = rotate_points(nxp, -angle, ego_locs[0]) + shift
        else:
            ego_locs = rotate_points(ego_locs, -angle, ego_locs[0]) + shift
            nxp = rotate_points(nxp, -angle, ego_locs[0]) + shift
            locs = rotate_points(locs, -angle, ego_locs[0]) + shift
            oris[1:] = oris[1:] - np.deg2rad(angle)
        p_locs, p_oris, p_typs, n = self._pad_actors(locs, oris, typs)
        return bev, -ego_locs, cmd, -nxp, bra, -p_locs, p_oris, p_typs, n


class TemporalBEVDataset(BEVDataset):
    """'temporal_bev' (train_bev_v2): the map is the 3 + 2 x 3 channel temporal stack (temporal_bev_dataset.py:12-101)."""
    temporal = True


# ---------------------------------------------------------------------------------------------------- LiDAR loaders
class LiDARDataset(RouteFrames):
    """'lidar': (lidar (max_points, 4), num_points, heatmaps, sizemaps, orimaps, bev, ego_locs, cmd, nxp, bra, locs, oris,
    typs, num_objs) (lidar_dataset.py:8-99)."""
    painted = False

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.x_edges = np.linspace(self.min_x, self.max_x, (self.max_x - self.min_x) * self.pixels_per_meter)
        self.y_edges = np.linspace(self.min_y, self.max_y, (self.max_y - self.min_y) * self.pixels_per_meter)
        self.margin = MARGIN
        if self.painted:
            yaws = self.camera_yaws[1:-1]
            self.cameras = [CameraProjection(y, [0, 0, self.camera_z], [self.camera_x, 0, self.camera_z], 288, 256, 64) for y in yaws]
            self.all_visible = np.ones((len(yaws), 1, 288, 256))

    def drop_ego_points(self, xyzr, painted=None):
        """Returns of the ego vehicle's own body (lidar_dataset.py:15-25)."""
        hit = (xyzr[:, 0] > -2.4) & (xyzr[:, 0] < 0) & (xyzr[:, 1] > -0.8) & (xyzr[:, 1] < 0.8) & (xyzr[:, 2] > -1.5) & (xyzr[:, 2] < -1)
        rows = np.argwhere(hit)
        if painted is None:
            return np.delete(xyzr, rows, axis=0)
        return np.delete(xyzr, rows, axis=0), np.delete(painted, rows, axis=0)

    preprocess = drop_ego_points

    def detections_to_heatmap(self, locs, oris, bbox, typs, radius=1):
        """Gaussian centre heat-maps per class plus size / orientation maps where a detection's Gaussian is the strongest so
        far (lidar_dataset.py:101-137)."""
        h, w = len(self.y_edges), len(self.x_edges)
        heat, size, ori_map = torch.zeros((2, h, w)), torch.zeros((2, h, w)), torch.zeros((2, h, w))
        for cls in (0, 1):
            sel = typs == cls
            if sum(sel) == 0:
                continue
            loc = torch.tensor(locs[sel], dtype=torch.float32)
            ori = torch.tensor(oris[sel], dtype=torch.float32)
            box = torch.tensor(bbox[sel], dtype=torch.float32)
            cx = -(loc[:, 0] * self.pixels_per_meter) + (self.max_y - self.min_y) * self.pixels_per_meter / 2
            cy = -(loc[:, 1] * self.pixels_per_meter) + h + self.min_x * self.pixels_per_meter
            gx = (-((torch.arange(w)[:, None] - cx[None, :]) / radius) ** 2).exp()
            gy = (-((torch.arange(h)[:, None] - cy[None, :]) / radius) ** 2).exp()
            g, who = (gx[None] * gy[:, None]).max(dim=-1)
            new = g > heat.max(dim=0)[0]
            size[:, new] = box.T[:, who[new]] * self.pixels_per_meter
            ori_map[0, new] = torch.from_numpy(np.cos(ori[who[new]].numpy()))
            ori_map[1, new] = torch.from_numpy(np.sin(ori[who[new]].numpy()))
            heat[cls] = g
        return heat, size, ori_map

    def __getitem__(self, idx):
        txn, index = self.txn_map[idx], self.idx_map[idx]
        xyzr = read_array(txn, "lidar", index).reshape(-1, 4)
        if self.painted:
            sem = read_array(txn, "lidar_sem", index).reshape(-1, len(self.seg_channels))
            xyzr, sem = self.drop_ego_points(xyzr, sem)
        _, ego_locs, ego_oris, _, _, locs, oris, bbox, typs = self._tracks(txn, index)
        ego_locs, locs, oris, bbox, typs = to_ego_frame(ego_locs, locs, oris, bbox, typs, ego_oris[0], self.num_plan + 1)
        angle = float(torch.rand(1) * 2 - 1) * self.angle_jitter
        cmd, bra, nxp = self._commands(txn, index)
        bev = (rotate_image(read_bev(txn, index, [0, 1, 2, 9, 10]), angle) > 0).astype(np.uint8).transpose(2, 0, 1)
        if not self.painted:
            xyzr = self.drop_ego_points(xyzr)
        xyzr = rotate_lidar(xyzr[:, :4], -angle)
        ego_locs = rotate_points(ego_locs, -angle, ego_locs[0])
        nxp = rotate_points(nxp, -angle, ego_locs[0])
        if self.painted:
            sem *= paint_from_cameras(xyzr, self.all_visible, self.cameras)
        locs = rotate_points(locs, -angle, ego_locs[0])
        oris[1:] = oris[1:] - np.deg2rad(angle)
        heat, size, ori_map = self.detections_to_heatmap(locs[:, 0], oris[:, 0], bbox[:, 0], typs[:, 0])
        p_locs, p_oris, p_typs, n = self._pad_actors(locs, oris, typs)
        order = np.arange(len(xyzr))
        np.random.shuffle(order)
        width = 4 + (len(self.seg_channels) if self.painted else 0)
        lidar = np.empty((self.max_lidar_points, width), np.float32)
        num_points = min(self.max_lidar_points, len(xyzr))
        lidar[:num_points, :4] = xyzr[order][:num_points]
        if self.painted:
            lidar[:num_points, 4:] = sem[order][:num_points]
        return lidar, num_points, heat, size, ori_map, bev, -ego_locs, cmd, -nxp, bra, -p_locs, p_oris, p_typs, n


class LiDARPaintedDataset(LiDARDataset):
    """'lidar_painted': points carry the recorded class scores of the pixels they project to, masked to the three cameras'
    fields of view after the rotation augment (lidar_painted_dataset.py:7-92)."""
    painted = True


class TemporalLiDARPaintedDataset(LiDARPaintedDataset):
    """'temporal_lidar_painted' (train_full_v2): current + num_frame_stack past sweeps in the current ego frame with a one-hot
    time channel, temporal BEV stack as the segmentation target, forecasting targets from the stricter
    max_mot_vehicle_radius (temporal_lidar_painted_dataset.py:12-176)."""

    def __getitem__(self, idx):
        txn, index = self.txn_map[idx], self.idx_map[idx]
        n_sem = len(self.seg_channels)
        angle = float(torch.rand(1) * 2 - 1) * self.angle_jitter
        sweeps = []
        for i in reversed(range(index - self.num_frame_stack, index + 1)):
            if i < 0:
                continue
            xyzr, sem = self.drop_ego_points(read_array(txn, "lidar", i).reshape(-1, 4), read_array(txn, "lidar_sem", i).reshape(-1, n_sem))
            _, locs_i, oris_i, *_ = self._tracks(txn, i)
            if i == index:
                loc0, ori0 = locs_i[0], oris_i[0]
                loc_jitter, ori_jitter = 0, 0
            else:
                loc_jitter = np.random.uniform(low=-self.stack_loc_jitter, high=self.stack_loc_jitter, size=2)
                ori_jitter = np.random.uniform(low=-self.stack_ori_jitter, high=self.stack_ori_jitter)
            xyzr = rotate_lidar(xyzr, -angle)
            sem *= paint_from_cameras(xyzr, self.all_visible, self.cameras)
            sweeps.append((move_lidar_points(xyzr, locs_i[0] - loc0 + loc_jitter, ori0, oris_i[0] + ori_jitter), sem))
        total = sum(len(x) for x, _ in sweeps)
        lidar = np.zeros((total, 4 + n_sem + self.num_frame_stack + 1), np.float32)
        at = 0
        for k, (xyzr, sem) in enumerate(sweeps):
            lidar[at:at + len(xyzr), :4] = xyzr
            lidar[at:at + len(xyzr), 4:4 + n_sem] = sem
            lidar[at:at + len(xyzr), 4 + n_sem + k] = 1.0
            at += len(xyzr)
        order = np.arange(len(lidar))
        np.random.shuffle(order)
        lidar = lidar[order[:self.max_lidar_points]]
        cmd, bra, nxp = self._commands(txn, index)
        # detection + segmentation targets (max_vehicle_radius)
        _, ego_locs, ego_oris, _, _, locs, oris, bbox, typs = self._tracks(txn, index)
        ego_locs, locs, oris, bbox, typs = to_ego_frame(ego_locs, locs, oris, bbox, typs, ego_oris[0], self.num_plan + 1)
        bev = self._bev_stack(txn, index, angle)
        locs = rotate_points(locs, -angle, ego_locs[0])
        oris[1:] = oris[1:] - np.deg2rad(angle)
        heat, size, ori_map = self.detections_to_heatmap(locs[:, 0], oris[:, 0], bbox[:, 0], typs[:, 0])
        p_locs, p_oris, p_typs, n = self._pad_actors(locs, oris, typs)
        padded = np.zeros((self.max_lidar_points, lidar.shape[1]), np.float32)
        num_points = min(self.max_lidar_points, total)
        padded[:num_points] = lidar[:num_points]
        # planning targets come from a second, stricter query (max_mot_vehicle_radius): only the ego track of it is used
        _, ego_locs, ego_oris, _, _, locs, oris, bbox, typs = self._tracks(txn, index, vehicle_radius=self.max_mot_vehicle_radius)
        ego_locs, *_ = to_ego_frame(ego_locs, locs, oris, bbox, typs, ego_oris[0], self.num_plan + 1)
        ego_locs = rotate_points(ego_locs, -angle, ego_locs[0])
        nxp = rotate_points(nxp, -angle, ego_locs[0])
        return padded, num_points, heat, size, ori_map, bev, -ego_locs, cmd, -nxp, bra, -p_locs, p_oris, p_typs, n


LOADERS = {"bev": BEVDataset, "temporal_bev": TemporalBEVDataset, "lidar": LiDARDataset, "lidar_painted": LiDARPaintedDataset,
           "temporal_lidar_painted": TemporalLiDARPaintedDataset}


def get_data_loader(data_type, args, rank: int = 0, world: int = 1):
    """lav/utils/datasets/__init__.py:12-40: shuffled, drop_last batches of `args.batch_size` from `args.config_path`'s data_dir.
    The reference feeds one loader of the global batch to nn.DataParallel; with one process per GPU (world > 1) every rank
    draws its own disjoint shard of each epoch (DistributedSampler; call loader.sampler.set_epoch(epoch)) in batches of
    batch_size / world."""
    if data_type not in LOADERS:
        raise NotImplementedError(f"data loader {data_type!r}: this build provides {sorted(LOADERS)} (the camera-model loaders "
                                  "'rgb', 'seg', 'bra' belong to trainers outside its scope)")
    dataset = LOADERS[data_type](args.config_path, seed=args.seed)
    common = dict(num_workers=args.num_workers, drop_last=True, pin_memory=torch.cuda.is_available())
    if world > 1:
        from torch.utils.data.distributed import DistributedSampler
        if args.batch_size % world:
            raise ValueError(f"global batch {args.batch_size} is not divisible by {world} ranks")
        sampler = DistributedSampler(dataset, num_replicas=world, rank=rank, shuffle=True, seed=args.seed, drop_last=True)
        return DataLoader(dataset, batch_size=args.batch_size // world, sampler=sampler, **common)
    return DataLoader(dataset, batch_size=args.batch_size, shuffle=True, **common)
