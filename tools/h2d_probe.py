"""Why does a pinned-host -> HBM Tensor.copy_(non_blocking=True) cost the HOST milliseconds inside the frame loop?  Host time of the call on an idle
GPU, behind a queue of kernels on the same stream, behind a graph replay, and on a second stream.    python tools/h2d_probe.py"""
import time

import torch

dev = torch.device("cuda:0")
src = torch.randn((3, 3, 288, 256)).pin_memory()
dst = torch.empty_like(src, device=dev)
a = torch.randn((4096, 4096), device=dev)


def host_us(fn, reps=20):
    torch.cuda.synchronize()
    t = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); t.append(time.perf_counter() - t0)
        torch.cuda.synchronize()
    t.sort()
    return t[len(t) // 2] * 1e6, t[-1] * 1e6


print("pinned:", src.is_pinned())
print("idle GPU:            median %.1f us, max %.1f us" % host_us(lambda: dst.copy_(src, non_blocking=True)))


def busy_then_copy():
    for _ in range(20):
        a @ a            # ~20 x 0.12 ms of queued work
    t0 = time.perf_counter()
    dst.copy_(src, non_blocking=True)
    return time.perf_counter() - t0


ts = []
for _ in range(20):
    torch.cuda.synchronize(); ts.append(busy_then_copy())
torch.cuda.synchronize(); ts.sort()
print("behind 20 queued matmuls (same stream): median %.1f us, max %.1f us" % (ts[10] * 1e6, ts[-1] * 1e6))
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(3):
        a @ a
torch.cuda.synchronize()
with torch.cuda.graph(g, stream=s):
    for _ in range(20):
        b = a @ a
torch.cuda.synchronize()
ts = []
for _ in range(20):
    torch.cuda.synchronize()
    g.replay()
    t0 = time.perf_counter(); dst.copy_(src, non_blocking=True); ts.append(time.perf_counter() - t0)
torch.cuda.synchronize(); ts.sort()
print("behind a graph replay (same stream):    median %.1f us, max %.1f us" % (ts[10] * 1e6, ts[-1] * 1e6))
s2 = torch.cuda.Stream()
ts = []
for _ in range(20):
    torch.cuda.synchronize()
    g.replay()
    with torch.cuda.stream(s2):
        t0 = time.perf_counter(); dst.copy_(src, non_blocking=True); ts.append(time.perf_counter() - t0)
torch.cuda.synchronize(); ts.sort()
print("other stream while a graph runs:        median %.1f us, max %.1f us" % (ts[10] * 1e6, ts[-1] * 1e6))
sl = src[:2]
print("slice of a pinned tensor pinned:", sl.is_pinned())
