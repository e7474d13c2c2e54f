import sys, time, torch
sys.path.insert(0, '/root/repo')
from lav_amd import ops
dev = torch.device("cuda")
feat = torch.randn((1, 384, 160, 160), device=dev)
for n in (1, 2, 4):
    locs = torch.tensor([[3.0 * i, -5.0 - i] for i in range(n)], device=dev)
    oris = torch.tensor([0.3 * i for i in range(n)], device=dev)
    for _ in range(5):
        ops.crop_rotate(feat, locs, oris, 2.0, 96, 0.0, 0.75)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        ops.crop_rotate(feat, locs, oris, 2.0, 96, 0.0, 0.75)
    e1.record(); torch.cuda.synchronize()
    print(f"crop n={n}: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us", flush=True)
