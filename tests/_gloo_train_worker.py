"""world_size-2 data-parallel train_bev on CPU (gloo): each rank trains on its own shard; after the step every rank
must hold identical parameters (the all-reduce averaged the gradients), different from what a rank would have got alone."""
import copy
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lav_amd.train import LAV, TrainConfig, synthetic_bev_batch  # noqa: E402

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
torch.set_num_threads(4)
cfg = TrainConfig()
lav = LAV(cfg, "cpu", what="bev")
alone = LAV(cfg, "cpu", what="bev")           # same seeded weights, no wrapper use: the single-rank control
alone.bev_ddp = alone.bev_planner
before = copy.deepcopy(lav.bev_planner.state_dict())
batch = synthetic_bev_batch(2, seed=31 + rank, num_objs=3)      # a different shard per rank
torch.manual_seed(5)
info = lav.train_bev(*batch, other_weight=0.5)
torch.manual_seed(5)
alone.train_bev(*batch, other_weight=0.5)
name = "plan_gru.weight_hh_l0"
mine = lav.bev_planner.state_dict()[name]
gathered = [torch.zeros_like(mine) for _ in range(world)]
dist.all_gather(gathered, mine)
assert all(torch.equal(gathered[0], g) for g in gathered), "ranks diverged: gradients were not all-reduced"
assert not torch.equal(mine, before[name]), "no update happened"
assert not torch.allclose(mine, alone.bev_planner.state_dict()[name], atol=1e-9), "update equals the single-rank update"
loss = torch.tensor([info["loss"]]); dist.all_reduce(loss)
if rank == 0:
    print("DDP_OK", round(float(loss) / world, 3))
dist.destroy_process_group()
