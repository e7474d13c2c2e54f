"""What each side graph costs the frame, and what stream priorities do (round 6): the bench's frame loop under in-process variants,
interleaved, `--rounds` times each.

    python tools/frame_ab.py [--steps 60] [--rounds 3]

variants: all (the headline frame) | no_brake | no_ego | chain (neither) | hp (the step issued on a high-priority stream: the main chain's
graphs replay there, the side streams stay at the default priority)."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from lav_amd import frame as frame_mod  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=60)
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--variants", default="all,no_brake,no_ego,chain,hp")
ap.add_argument("--graphs", action="store_true", help="also the stand-alone replay time of every graph")
a = ap.parse_args()
device = torch.device("cuda", 0)
pipe, sds, _ = bench.build_pipeline(device)
host, dev = bench.synthetic_inputs(device)
nt = len(dev["ticks"])
i = 0


def step():
    global i
    loc, ori = bench.pose(i)
    out = pipe.step(dev["ticks"][i % nt], dev["all_rgbs"], dev["rgbs"], dev["tel_rgbs"], loc, ori, dev["nxp"], 3)
    i += 1
    return out


pipe.precapture(cmds=[3], max_others=8)
for _ in range(20):
    step()
torch.cuda.synchronize()
hp = torch.cuda.Stream(device, priority=-1)
SKIP = {"all": set(), "no_brake": {"brake"}, "no_ego": {"ego"}, "chain": {"brake", "ego"}, "hp": set()}
res = {v: [] for v in a.variants.split(",")}
FORCED = {"n0": 0, "n4": 4}     # the others branch forced to that many fixed poses (bench.py's forced_others)
for _ in range(a.rounds):
    for v in res:
        if v in FORCED:
            k = FORCED[v]
            pipe.set_forced_others([[4.0 + 3.0 * j, -8.0 - 4.0 * j] for j in range(k)], [0.2 * j - 0.3 for j in range(k)])
        frame_mod._DIAG_SKIP = SKIP.get(v, set())
        ctx = torch.cuda.stream(hp) if v == "hp" else torch.cuda.stream(torch.cuda.current_stream())
        with ctx:
            for _ in range(6):
                step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.steps):
                step()
            torch.cuda.synchronize()
            res[v].append(round((time.perf_counter() - t0) / a.steps * 1e3, 4))
        if v in FORCED:
            pipe.set_forced_others(None)
frame_mod._DIAG_SKIP = set()


def graph_times(iters=200):
    """Stand-alone replay time of each frame graph, us (nothing else on the GPU)."""
    out = {}
    for key, g in pipe.graphs.items():
        name = key if isinstance(key, str) else "_".join(str(k) for k in key)
        state = (pipe.ring.clone(), pipe.b_prev.clone())
        torch.cuda.synchronize()
        for _ in range(5):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            g.replay()
        torch.cuda.synchronize()
        out[name] = round((time.perf_counter() - t0) / iters * 1e6, 1)
        pipe.ring.copy_(state[0]); pipe.b_prev.copy_(state[1])
    return out


if a.graphs:
    print(json.dumps(dict(graphs_us=graph_times())))
print(json.dumps({v: dict(ms=r, best=min(r), fps=round(1e3 / min(r), 1)) for v, r in res.items()}))
