#!/usr/bin/env python3
"""Would Winograd F(2x2, 3x3) pay for the head convolution (384 -> 256 channels, 160 x 160, batch 1; direct MFMA kernel: 467 us)?
Its core is 16 independent GEMMs U_f (256 x 384) . V_f (384 x 6400): timed here as one rocBLAS batched GEMM (fp32), next to the
bytes the two transform passes would have to move."""
import torch

dev = torch.device("cuda:0")


def timeit(fn, reps=30):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for cout, cin, tiles, name in ((256, 384, 6400, "heads 384->256 @160x160"), (64, 64, 6400, "bev 64->64 @160x160"), (128, 128, 1600, "bev 128->128 @80x80")):
    U = torch.randn(16, cout, cin, device=dev)
    V = torch.randn(16, cin, tiles, device=dev)
    out = torch.empty(16, cout, tiles, device=dev)
    t = timeit(lambda: torch.bmm(U, V, out=out))
    fl = 2.0 * 16 * cout * cin * tiles
    direct = 2.0 * cout * cin * 9 * tiles * 4
    mb = (16 * cin * tiles + 16 * cout * tiles) * 4 * 2 / 1e6
    print(f"{name:26s}: batched GEMM {t:7.1f} us = {fl / t / 1e6:6.1f} TFLOP/s ({direct / t / 1e6:6.1f} TFLOP/s of direct-convolution flops); "
          f"transform traffic {mb:6.1f} MB (~{mb / 4.5e3 * 1e3:5.1f} us at 4.5 TB/s)", flush=True)
