"""Where does the HOST spend a frame?  GraphedFramePipeline.step with LAV_FRAME_HOST_TRACE=1 stamps time.perf_counter() at its phases; this
tool runs bench.py's frame loop (inputs resident in HBM, frames strictly one after the other) and prints the mean time between stamps,
including the gap between one step's return and the next step's entry (the caller's loop) and the frame period.  The GPU can only start
frame t+1's first graph once the host has launched it: whatever the host does between "heads done on the GPU" and "lidar graph launched"
of the next frame that is not covered by the others branch still running on the GPU is exposed frame time.

    python tools/host_tail_probe.py [frames]"""
import os
import sys
import time

os.environ["LAV_FRAME_HOST_TRACE"] = "1"
import torch  # noqa: E402

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device("cuda:0")
pipe, sds, _ = bench.build_pipeline(dev)
host, d = bench.synthetic_inputs(dev)
nt = len(d["ticks"])
pipe.precapture(cmds=[3], max_others=8)
if os.environ.get("FORCED_OTHERS") is not None:
    n_f = int(os.environ["FORCED_OTHERS"])
    pipe.set_forced_others(torch.tensor([[3.0 * k, -5.0 - 2 * k] for k in range(n_f)], device=dev).reshape(n_f, 2), torch.tensor([0.3 * k for k in range(n_f)], device=dev))


if os.environ.get("HOST_INPUTS"):   # the sensor tensors come from pinned host memory (bench.py's with_sensor_upload block)
    d = dict(ticks=[torch.from_numpy(t).pin_memory() for t in host["ticks"]], all_rgbs=torch.from_numpy(host["all_rgbs"]).pin_memory(),
             rgbs=torch.from_numpy(host["rgbs"]).pin_memory(), tel_rgbs=torch.from_numpy(host["tel_rgbs"]).pin_memory(),
             nxp=torch.from_numpy(host["nxp"]).pin_memory())


def step(i):
    loc, ori = bench.pose(i)
    return pipe.step(d["ticks"][i % nt], d["all_rgbs"], d["rgbs"], d["tel_rgbs"], loc, ori, d["nxp"], 3)


i = 0
for _ in range(24):
    step(i); i += 1
torch.cuda.synchronize()
pipe.host_trace.clear()
t0 = time.perf_counter()
for _ in range(N):
    step(i); i += 1
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / N
tr = pipe.host_trace
per = {}
order = []
for (l0, a), (l1, b) in zip(tr[:-1], tr[1:]):
    key = f"{l0}  ->  {l1}"
    if key not in per:
        per[key] = []; order.append(key)
    per[key].append(b - a)
print(f"{torch.cuda.get_device_name(0)}: frame period {dt * 1e3:.3f} ms ({1 / dt:.1f} frames/s) over {N} frames, FORCED_OTHERS={os.environ.get('FORCED_OTHERS')}")
tot = 0.0
for k in order:
    m = sum(per[k]) / len(per[k]) * 1e6
    tot += m
    print(f"  {m:8.1f} us  {k}")
print(f"  {tot:8.1f} us  sum (= the frame period when the host never idles outside the event wait)")
