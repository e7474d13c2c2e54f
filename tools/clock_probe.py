"""What clock and power does the chip run at under the head convolution?  The 384 -> 256 3x3 convolution (bf16x6, k_conv_split<2,2>) is
launched back to back for a few seconds while rocm-smi is polled; the same for the stream-K launch and for an idle chip.  The roofline
fractions in bench.py are against the 2.4 GHz peak of /opt/skills/guides/MI355X_MICROARCH.md.    python tools/clock_probe.py"""
import os
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lav_amd.ops import ConvLayer  # noqa: E402

dev = torch.device("cuda")
x = torch.randn((1, 384, 160, 160), device=dev)
w = torch.randn((256, 384, 3, 3)) / (384 * 9) ** 0.5
layer = ConvLayer(w, padding=1, device=dev)


def poll(stop, out):
    while not stop.is_set():
        try:
            r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp"], capture_output=True, text=True, timeout=5).stdout
            out.append([ln.strip() for ln in r.splitlines() if any(k in ln for k in ("sclk", "mclk", "fclk", "Power", "junction", "Temperature (Sensor edge)"))])
        except Exception as e:  # noqa: BLE001
            out.append([repr(e)])
        time.sleep(0.4)


def run(label, seconds, fn):
    stop, out = threading.Event(), []
    t = threading.Thread(target=poll, args=(stop, out)); t.start()
    t0 = time.time(); n = 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.time() - t0 < seconds:
        for _ in range(50):
            fn()
        n += 50
        torch.cuda.synchronize()
    e1.record(); torch.cuda.synchronize()
    stop.set(); t.join()
    per = e0.elapsed_time(e1) / max(n, 1) * 1e3 if n else 0.0
    print(f"== {label}: {n} launches, {per:.1f} us each (incl. host gaps)")
    for s in out[1:6]:
        print("   ", " | ".join(s))


run("idle", 2.0, lambda: None)
run(f"head convolution (LAV_SPLIT_SK={os.environ.get('LAV_SPLIT_SK', '1')})", 6.0, lambda: layer(x))
a = torch.randn((8192, 8192), device=dev, dtype=torch.bfloat16)
run("torch bf16 8192^3 matmul (hipBLASLt)", 6.0, lambda: a @ a)
