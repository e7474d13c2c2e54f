"""In-kernel cycle shares of the split-operand convolution (LAV_SPLIT_TRACE=1: prologue wait, loop, barrier waits of the compute and loader
waves) on the BEV backbone's layer shapes at LAV_CONV_F16X3, with producer maxima (the frame's configuration).

    LAV_SPLIT_TRACE=1 python tools/split_trace.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lav_amd import _lib
from lav_amd.ops import Amax, ConvLayer

dev = torch.device("cuda", 0)
SHAPES = [("bev 64->64 s2 320", 64, 64, 2, 320), ("bev 64->64 160", 64, 64, 1, 160), ("bev 64->128 s2 160", 64, 128, 2, 160),
          ("bev 128->128 80", 128, 128, 1, 80)]
for name, cin, cout, s, hw in SHAPES:
    torch.manual_seed(0)
    layer = ConvLayer(torch.randn(cout, cin, 3, 3) / (cin * 9) ** 0.5, stride=s, padding=(1, 1), relu_pre=True,
                      bn=(torch.zeros(cout), torch.ones(cout), torch.ones(cout), torch.zeros(cout)), precision=_lib.CONV_F16X3, device=dev)
    x = torch.randn(1, cin, hw, hw, device=dev).relu_()
    am_in, am_out = Amax(dev), Amax(dev)
    am_in.take(1).fill_(float(x.abs().max()))
    print("==", name, file=sys.stderr, flush=True)
    for _ in range(16):
        am_out.reset()
        y = layer(x, amax_in=am_in, amax_out=am_out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    os.environ.pop("LAV_SPLIT_TRACE", None)
    e0.record()
    for _ in range(50):
        am_out.reset()
        y = layer(x, amax_in=am_in, amax_out=am_out)
    e1.record()
    torch.cuda.synchronize()
    print(f"   {e0.elapsed_time(e1) / 50 * 1e3:.1f} us per launch (back to back, eager)", file=sys.stderr, flush=True)
