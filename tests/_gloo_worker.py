import time

import torch
import torch.distributed as dist

dist.init_process_group("gloo")
r = dist.get_rank()
dist.barrier()
t0 = time.perf_counter()
time.sleep(0.05 * (r + 1))
dt = time.perf_counter() - t0
t = torch.tensor([dt], dtype=torch.float64)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
assert t.item() >= 0.1 - 1e-3, t
if r == 0:
    print("MAX_OK", round(t.item(), 2))
dist.destroy_process_group()
