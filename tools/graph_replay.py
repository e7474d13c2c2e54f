"""Replay ONE graph of the frame by itself (under rocprofv3 --kernel-trace: a single stream, so the trace's serialisation of streams does
not distort it): what every launch of the lidar / heads / ego / others graph takes in its own chain.

    rocprofv3 --kernel-trace --output-format csv -d /tmp/gt -- python tools/graph_replay.py lidar 30
    python tools/trace_slice.py /tmp/gt 20 21        (the lidar graph: one replay = one k_merge_ticks)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "lidar"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
device = torch.device("cuda", 0)
pipe, sds, _ = bench.build_pipeline(device)
host, dev = bench.synthetic_inputs(device)
nt = len(dev["ticks"])
pipe.precapture(cmds=[3], max_others=8)
for i in range(20):
    loc, ori = bench.pose(i)
    pipe.step(dev["ticks"][i % nt], dev["all_rgbs"], dev["rgbs"], dev["tel_rgbs"], loc, ori, dev["nxp"], 3)
torch.cuda.synchronize()
key = {"lidar": "lidar", "heads": "heads", "brake": "brake", "others": "others_cap", "ego": ("ego", 3)}[which]
if key == "brake" and "brake" not in pipe.graphs:     # (default since round 6: the brake net is two graphs, replayed one after the other here)
    class _Both:
        def replay(self):
            pipe.graphs["brake_a"].replay(); pipe.graphs["brake_b"].replay()
    g = _Both()
else:
    g = pipe.graphs[key]
state = (pipe.ring.clone(), pipe.b_prev.clone())
for _ in range(n):
    g.replay()
    torch.cuda.synchronize()
pipe.ring.copy_(state[0]); pipe.b_prev.copy_(state[1])
print("replayed", which, n)
