"""world_size-2 data-parallel train_lidar on CPU (gloo): the full student (_Student = LiDARModel + UniPlanner) behind
DistributedDataParallel, with a shard that has NO eligible vehicle on rank 1 (UniPlanner.forward's `pick is None` branch:
cast_cmd_pred gets no gradient there) - the case that deadlocks DDP unless unused parameters are handled.

The HIP front end of the training graph does not exist on a CPU box, so this worker - TEST INFRASTRUCTURE - swaps in
torch stand-ins for the two liblav_amd training ops (pillar_decorate, scatter_max) and keeps the frozen teacher on its
torch (train-mode) code path.  What is under test is the data-parallel wiring, not numerics."""
import copy
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lav_amd import ops  # noqa: E402
from lav_amd.bev_planner import BEVPlanner  # noqa: E402
from lav_amd.train import LAV, TrainConfig, synthetic_lidar_batch  # noqa: E402


def pillar_decorate_cpu(points, num_points, grid):
    """grid_locations + unique + scatter_mean + decorate (lav/models/point_pillar.py:55-85) with torch ops."""
    if points.dim() == 2:
        points = points[None]
    pts, coords, src = [], [], []
    nmax = points.shape[1]
    for b in range(points.shape[0]):
        p = points[b, : int(num_points[b])]
        keep = (p[:, 0] >= grid.min_x) & (p[:, 0] < grid.max_x) & (p[:, 1] >= grid.min_y) & (p[:, 1] < grid.max_y)
        idx = torch.nonzero(keep)[:, 0]
        p = p[idx]
        c = ((p[:, :2] - torch.tensor([grid.min_x, grid.min_y])) * grid.ppm).long()
        coords.append(torch.cat([torch.full((len(c), 1), b), c], 1)); pts.append(p); src.append(idx + b * nmax)
    pts, coords, src = torch.cat(pts), torch.cat(coords), torch.cat(src)
    uc, inv = coords.unique(return_inverse=True, dim=0)
    cnt = torch.zeros(len(uc)).index_add_(0, inv, torch.ones(len(inv)))
    mean = torch.zeros((len(uc), 3)).index_add_(0, inv, pts[:, :3]) / cnt[:, None]
    xc = uc[inv][:, 2:3].float() / grid.ppm + grid.min_x
    yc = uc[inv][:, 1:2].float() / grid.ppm + grid.min_y
    dec = torch.cat([pts, pts[:, :3] - mean[inv], pts[:, :1] - xc, pts[:, 1:2] - yc], -1)
    return dec, uc.int(), inv.int(), src.int()


def scatter_max_cpu(src, index, num_segments):
    out = torch.zeros((num_segments, src.shape[1])).scatter_reduce(0, index.long()[:, None].expand_as(src), src, "amax", include_self=False)
    return out, None


ops.pillar_decorate, ops.scatter_max = pillar_decorate_cpu, scatter_max_cpu
BEVPlanner.eval = lambda self: self          # the teacher stays on its torch code path (its inference kernels are HIP)

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
torch.set_num_threads(4)
cfg = TrainConfig(log_inference=False)
lav = LAV(cfg, "cpu", what="lidar")
lav.bev_planner.train()
before = copy.deepcopy(lav.uniplanner.cast_cmd_pred.state_dict())
batch = list(synthetic_lidar_batch(1, seed=51 + rank, max_points=6000, num_objs=3))
if rank == 1:
    batch[12] = torch.zeros_like(batch[12])          # typs: no vehicle on this rank's shard
for step in range(2):                                # the second step would hang if the first left a bucket unreduced
    torch.manual_seed(7 + step)
    info = lav.train_lidar(*batch)
for name in ("uniplanner.cast_cmd_pred.0.weight", "uniplanner.plan_gru.weight_hh_l0", "lidar_model.backbone.conv1.0.weight"):
    mine = lav.student.state_dict()[name]
    gathered = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(gathered, mine)
    assert all(torch.equal(gathered[0], g) for g in gathered), f"ranks diverged on {name}: gradients were not all-reduced"
assert not torch.equal(lav.uniplanner.cast_cmd_pred.state_dict()["0.weight"], before["0.weight"]), \
    "cast_cmd_pred took no update although rank 0's shard trains it"
loss = torch.tensor([info["loss"]]); dist.all_reduce(loss)
if rank == 0:
    print("DDP_LIDAR_OK", round(float(loss) / world, 3))
dist.destroy_process_group()
