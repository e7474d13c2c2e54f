#!/usr/bin/env python3
"""Do two linear HIP graphs on two streams overlap?  Times lidar / brake graphs alone and together. GPU only."""
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402

dev = torch.device("cuda:0")
pipe, sds, _ = bench.build_pipeline(dev)
host, d = bench.synthetic_inputs(dev)
for i in range(25):
    loc, ori = bench.pose(i)
    pipe.step(d["ticks"][i % 4], d["all_rgbs"], d["rgbs"], d["tel_rgbs"], loc, ori, d["nxp"], 3)
torch.cuda.synchronize()
g = pipe.graphs
sA, sB, sC = torch.cuda.Stream(dev), torch.cuda.Stream(dev), torch.cuda.Stream(dev)


def run(label, plan, iters=50):
    """plan: list of (stream, graph key)"""
    for _ in range(3):
        for s, k in plan:
            with torch.cuda.stream(s):
                g[k].replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        for s, k in plan:
            with torch.cuda.stream(s):
                g[k].replay()
        torch.cuda.synchronize()
    print(f"{label:44s} {(time.perf_counter() - t0) / iters * 1e3:7.3f} ms", flush=True)


dflt = torch.cuda.default_stream(dev)
run("lidar alone (side stream)", [(sA, "lidar")])
run("lidar alone (default stream)", [(dflt, "lidar")])
run("brake alone", [(sB, "brake")])
run("heads alone", [(sA, "heads")])
run("ego alone", [(sC, ("ego", 3))])
run("lidar + brake, two side streams", [(sA, "lidar"), (sB, "brake")])
run("lidar(default stream) + brake(side)", [(dflt, "lidar"), (sB, "brake")])
run("brake + brake(same graph twice, 2 streams)", [(sA, "brake"), (sB, "brake")])
run("heads + ego, two side streams", [(sA, "heads"), (sC, ("ego", 3))])
run("heads + ego + brake, three streams", [(sA, "heads"), (sC, ("ego", 3)), (sB, "brake")])
run("lidar + brake + ego, three streams", [(sA, "lidar"), (sB, "brake"), (sC, ("ego", 3))])
