"""TEST INFRASTRUCTURE ONLY (bench.py's cpu_baseline leg and tests/): the train-mode forward of PointPillarNet on the host, so
that the train_full_v2 step can be timed on CPU cores.  The product's train-mode forward (lav_amd/point_pillar.py:forward_train)
does its index work with liblav_amd (lav_pillar_decorate, lav_scatter_max) and has no CPU path.

Restates lav/models/point_pillar.py:92-116: grid_locations + pillar_generation + decorate under no_grad (numpy: oracle/pillar.py,
pinned by tests/golden/pillar.npz), the PointNet = the module's own Linear / BatchNorm1d / ReLU on batch statistics,
torch_scatter.scatter_max as index_reduce('amax') (empty segments 0, as torch_scatter), scatter_points as an index assignment.
"""
import numpy as np
import torch

from . import pillar as opillar


def pillar_forward_train(ppn, lidar_list, num_points):
    """ppn: lav_amd.PointPillarNet in train mode on the CPU; lidar_list: list of (Ni, D) tensors or a (B, Nmax, D) tensor."""
    if torch.is_tensor(num_points):
        num_points = num_points.tolist()
    clouds = [np.asarray(lidar_list[b][: int(num_points[b])].detach().cpu().numpy(), np.float32) for b in range(len(num_points))]
    pts, coords = [], []
    for b, c in enumerate(clouds):
        kept, xy, _ = opillar.grid_locations(c, ppn.min_x, ppn.max_x, ppn.min_y, ppn.max_y, ppn.pixels_per_meter)
        pts.append(kept)
        coords.append(np.concatenate([np.full((len(kept), 1), b, xy.dtype), xy], axis=1))
    pts, coords = np.concatenate(pts), np.concatenate(coords)
    uniq, inv = opillar.pillar_generation(coords)
    dec = opillar.decorate(pts, uniq, inv, ppn.min_x, ppn.min_y, ppn.pixels_per_meter)
    feat = ppn.point_net.net(torch.from_numpy(np.ascontiguousarray(dec, np.float32)))
    index = torch.from_numpy(np.asarray(inv, np.int64))
    fmax = torch.full((len(uniq), feat.shape[1]), float("-inf"), dtype=feat.dtype).index_reduce(0, index, feat, "amax", include_self=True)
    fmax = torch.where(torch.isinf(fmax), torch.zeros_like(fmax), fmax)
    uc = torch.from_numpy(np.asarray(uniq, np.int64))
    ny, nx = int(ppn.ny), int(ppn.nx)
    canvas = torch.zeros((len(clouds), feat.shape[1], ny, nx), dtype=feat.dtype)
    canvas[uc[:, 0], :, torch.clamp(ny - 1 - uc[:, 1], 0, ny - 1), torch.clamp(uc[:, 2], 0, nx - 1)] = fmax
    return canvas


class teacher_on_cpu:
    """Context manager: the frozen privileged BEVPlanner of the train_lidar step evaluated with torch ops on the host.  In the
    product it runs on the HIP inference kernels (eval mode has no CPU path); here its eval-mode forward is assembled from the
    modules' own torch code paths - grid_sample crops, the ResNet-18 trunk with BatchNorm on running statistics, the torch GRU
    decoders - which is what the reference's teacher does on a CPU (lav/models/uniplanner.py:56-150)."""

    def __init__(self, teacher):
        self.t = teacher

    def __enter__(self):
        from lav_amd.planner_common import crop_feature_torch
        t = self.t
        trunk = t.bev_conv_emb[0]

        def crop(features, rel_locs, rel_oris, pixels_per_meter=4, crop_size=96, map_index=None):
            if map_index is not None:
                features = features[map_index.long()]
            return crop_feature_torch(features, rel_locs, rel_oris, pixels_per_meter, crop_size, *t.offsets())

        def plan(embd, nxp, cast_locs=None, pixels_per_meter=4, crop_size=96, cmd=-1, impl="auto"):
            if cast_locs is None:
                cast_locs = t._cast_torch(embd)
            return t._plan_torch(embd, nxp, cast_locs.detach(), pixels_per_meter, crop_size, cmd)

        t.__dict__["crop_feature"], t.__dict__["cast"], t.__dict__["plan"] = crop, (lambda embd, mode="ego": t._cast_torch(embd)), plan
        trunk.__dict__["forward"] = trunk.forward_train          # BatchNorm modules are in eval mode: running statistics
        return self

    def __exit__(self, *exc):
        for k in ("crop_feature", "cast", "plan"):
            self.t.__dict__.pop(k, None)
        self.t.bev_conv_emb[0].__dict__.pop("forward", None)
        return False
