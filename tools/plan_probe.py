"""Stand-alone time of the persistent plan launch (graph replay, nothing else on the GPU) and of the whole frame.
    LAV_PLAN_POLL=all|quarter python tools/plan_probe.py"""
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402
from lav_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
pipe, sds, (lm, up, seg, bra) = bench.build_pipeline(dev)
host, d = bench.synthetic_inputs(dev)
pipe.precapture(cmds=[3])
i = 0
for _ in range(20):
    loc, ori = bench.pose(i)
    pipe.step(d["ticks"][i % 4], d["all_rgbs"], d["rgbs"], d["tel_rgbs"], loc, ori, d["nxp"], 3); i += 1
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(60):
    loc, ori = bench.pose(i)
    out = pipe.step(d["ticks"][i % 4], d["all_rgbs"], d["rgbs"], d["tel_rgbs"], loc, ori, d["nxp"], 3); i += 1
torch.cuda.synchronize()
frame_ms = (time.perf_counter() - t0) / 60 * 1e3
h = pipe.health()
embd = out["ego_embd"].clone()
cast = up.cast(embd, mode="ego")
fn = lambda: up.plan(embd, pipe.b_nxp[None], cast_locs=cast, pixels_per_meter=up.pixels_per_meter, crop_size=up.crop_size * 2, cmd=3)
ref = up.plan(embd, pipe.b_nxp[None], cast_locs=cast, pixels_per_meter=up.pixels_per_meter, crop_size=up.crop_size * 2, cmd=3, impl="steps")
got = fn()
torch.cuda.synchronize()
err = float((got - ref).abs().max())
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    fn()
for _ in range(5):
    g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(100):
    g.replay()
torch.cuda.synchronize()
plan_ms = (time.perf_counter() - t0) / 100 * 1e3
print(f"LAV_PLAN_IMPL={os.environ.get('LAV_PLAN_IMPL', 'default (wave)')} LAV_PLAN_POLL={os.environ.get('LAV_PLAN_POLL', 'default')}: plan launch {plan_ms:.3f} ms alone, frame {frame_ms:.3f} ms = {1e3 / frame_ms:.1f} frames/s, "
      f"persistent vs steps |diff| {err:.2e}, plan_aborts {h['plan_aborts']}/{h['plan_launches']}, nonfinite {h['nonfinite_outputs']}, "
      f"finite plan {bool(torch.isfinite(out['ego_plan_locs']).all())}")
