"""Frame period with the tick's sensor tensors coming from the HOST, by the way they are uploaded (bench.py's with_sensor_upload block):
    resident   inputs already in HBM (the headline number's configuration)
    to         torch.from_numpy(x).to(device) per tensor (pageable, blocking), then step() on device tensors
    staged     pinned host tensors -> preallocated device tensors with copy_(non_blocking=True) on the current stream, then step()
    direct     step() handed the pinned host tensors (it copies them into the graphs' static buffers itself)
    python tools/upload_probe.py [frames]"""
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
dev = torch.device("cuda:0")
pipe, sds, _ = bench.build_pipeline(dev)
host, d = bench.synthetic_inputs(dev)
nt = len(d["ticks"])
pipe.precapture(cmds=[3], max_others=8)
keys = ("all_rgbs", "rgbs", "tel_rgbs", "nxp")
pin = dict(ticks=[torch.from_numpy(t).pin_memory() for t in host["ticks"]], **{k: torch.from_numpy(host[k]).pin_memory() for k in keys})
stage = dict(tick=torch.empty_like(d["ticks"][0]), **{k: torch.empty_like(d[k]) for k in keys})
i = 0


def frame(mode):
    global i
    loc, ori = bench.pose(i)
    j = i % nt
    if mode == "resident":
        args = (d["ticks"][j], d["all_rgbs"], d["rgbs"], d["tel_rgbs"])
        nxp = d["nxp"]
    elif mode == "to":
        args = tuple(torch.from_numpy(x).to(dev) for x in (host["ticks"][j], host["all_rgbs"], host["rgbs"], host["tel_rgbs"]))
        nxp = torch.from_numpy(host["nxp"]).to(dev)
    elif mode == "staged":
        stage["tick"].copy_(pin["ticks"][j], non_blocking=True)
        for k in keys:
            stage[k].copy_(pin[k], non_blocking=True)
        args = (stage["tick"], stage["all_rgbs"], stage["rgbs"], stage["tel_rgbs"])
        nxp = stage["nxp"]
    else:
        args = (pin["ticks"][j], pin["all_rgbs"], pin["rgbs"], pin["tel_rgbs"])
        nxp = pin["nxp"]
    out = pipe.step(*args, loc, ori, nxp, 3)
    i += 1
    return out


for _ in range(24):
    frame("resident")
for mode in ("resident", "to", "staged", "direct", "resident"):
    for _ in range(6):
        frame(mode)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N):
        frame(mode)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / N
    print(f"{mode:9s} {dt * 1e3:8.3f} ms per frame  {1 / dt:7.1f} frames/s", flush=True)
