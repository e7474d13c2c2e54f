"""When each graph of the frame starts and ends on the GPU (round 6): timing events recorded on the replaying stream right before and
after every graph replay of `GraphedFramePipeline.step`, relative to an event at the head of the step.  The extra event packets cost a
few us each - this is a map of the frame, not a benchmark.

    python tools/frame_timeline.py [--steps 40] [--forced N]"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=40)
ap.add_argument("--forced", type=int, default=-1, help="force the others branch to that many fixed poses (-1: as detected)")
a = ap.parse_args()
device = torch.device("cuda", 0)
pipe, sds, _ = bench.build_pipeline(device)
host, dev = bench.synthetic_inputs(device)
nt = len(dev["ticks"])
pipe.precapture(cmds=[3], max_others=8)
if a.forced >= 0:
    k = a.forced
    pipe.set_forced_others([[4.0 + 3.0 * j, -8.0 - 4.0 * j] for j in range(k)], [0.2 * j - 0.3 for j in range(k)])
i = 0


def step():
    global i
    loc, ori = bench.pose(i)
    out = pipe.step(dev["ticks"][i % nt], dev["all_rgbs"], dev["rgbs"], dev["tel_rgbs"], loc, ori, dev["nxp"], 3)
    i += 1
    return out


for _ in range(20):
    step()
torch.cuda.synchronize()
marks = []
orig = pipe._replay


def traced(key, fn, stream, *args, **kw):
    name = key if isinstance(key, str) else "_".join(str(k) for k in key)
    s = torch.cuda.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    out = orig(key, fn, stream, *args, **kw)
    e1.record(s)
    marks.append((name, e0, e1))
    return out


pipe._replay = traced
rows = {}
for _ in range(a.steps):
    marks.clear()
    t0 = torch.cuda.Event(enable_timing=True)
    t0.record(torch.cuda.current_stream())
    step()
    tend = torch.cuda.Event(enable_timing=True)
    tend.record(torch.cuda.current_stream())
    torch.cuda.synchronize()
    for name, e0, e1 in marks:
        rows.setdefault(name, []).append((t0.elapsed_time(e0), t0.elapsed_time(e1)))
    rows.setdefault("frame_end", []).append((t0.elapsed_time(tend), t0.elapsed_time(tend)))
out = {}
for name, v in rows.items():
    v = np.asarray(v)
    out[name] = dict(start_ms=round(float(np.median(v[:, 0])), 3), end_ms=round(float(np.median(v[:, 1])), 3),
                     dur_ms=round(float(np.median(v[:, 1] - v[:, 0])), 3))
print(json.dumps(out))
