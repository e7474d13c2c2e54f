"""Which instruction patterns go wrong beside a matrix + LDS heavy neighbour?  (VERDICT r4 #1 "try to explain")

Runs the minimal victims of tools/probes/lds_hazard_probe.hip (one hand-written LDS instruction sequence each, every one a sequence
hipcc may emit for plain C++) alone and beside tools/probes/lds_hog.hip in its four modes, and prints a table
pattern x neighbour -> wrong results / checks.

    python tools/lds_hazard.py [rounds] > profiles/r05_lds_hazard.txt
"""
import ctypes
import os
import subprocess
import sys

import torch

here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "probes")


def build(name):
    so = os.path.join(here, f"lib{name}.so")
    src = os.path.join(here, f"{name}.hip")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", src, "-o", so])
    return ctypes.CDLL(so)


hog = build("lds_hog")
haz = build("lds_hazard_probe")
hog.hog_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
haz.hazard_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]

dev = torch.device("cuda")
ROUNDS = int(sys.argv[1]) if len(sys.argv) > 1 else 30
PATTERNS = {
    0: "control: write, wait, read, wait",
    1: "ds_write_b32 a, v ; v_mov v, junk           (store data WAR)",
    2: "ds_write_b32 a, v ; v_mov a, other          (store address WAR)",
    3: "ds_read_b32 r, a ; v_mov a, other           (load address WAR)",
    4: "2 loads in flight, registers pinned, lgkmcnt(0)",
    5: "8 loads in flight, registers pinned, lgkmcnt(0)",
    6: "2 loads, s_waitcnt lgkmcnt(1), use the first (in-order return)",
    7: "store then load of the same word, back to back",
    8: "store A + load B in flight, registers pinned",
    9: "ds_write_b64 ; v_mov_b64 data, junk",
    10: "ds_write2_b64 ; v_mov_b64 data1, junk        (round 4's compiler output)",
    11: "ds_write2_b64 ; s_nop 0 ; v_mov_b64 data1, junk",
    12: "ds_bpermute_b32 + ds_read_b32 in flight, registers pinned",
    13: "C++: 6 interleaved __shfl_xor butterflies vs the same sums on permlane/DPP",
    14: "C++: 13 + lane-0 stores, LDS barrier, read back (the LDS plan kernel's step)",
    15: "NO LDS: a chain of vector-ALU set-up arithmetic evaluated twice, bits compared",
}
PER_ITER = {13: 6, 14: 6, 15: 1, 0: 1, 1: 1, 2: 2, 3: 1, 4: 2, 5: 8, 6: 2, 7: 1, 8: 2, 9: 2, 10: 4, 11: 4, 12: 2}
NEIGHBOURS = [("alone", None), ("matrix only (hog 0)", 0), ("matrix + LDS (hog 1)", 1), ("LDS only (hog 3)", 3), ("ERFNet 16-ch pair run", "chain16")]
# the frame's own strongest neighbour: ERFNet's 16-channel persistent pair run (432 row workgroups x 256 threads, 66-80 KB of LDS each:
# the only matrix + LDS kernel of the frame that leaves tens of KB of LDS free on its CUs) - tools/crop_victim.py
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch.nn as nn  # noqa: E402
from lav_amd.ops import Conv1dPair, Conv1dPairChain  # noqa: E402
_pairs = [Conv1dPair(nn.Conv2d(16, 16, (3, 1), padding=(1, 0)), nn.Conv2d(16, 16, (1, 3), padding=(0, 1)), nn.BatchNorm2d(16, eps=1e-3).eval(), device="cuda") for _ in range(10)]
_chain = Conv1dPairChain(_pairs, [i % 2 == 1 for i in range(10)])
_cx = torch.randn((3, 16, 144, 128), device="cuda")
s_hog, s_vic = torch.cuda.Stream(), torch.cuda.Stream()
sink = torch.zeros(16, device=dev)
print(f"# {torch.cuda.get_device_name(0)}; victim: 2048 workgroups x 256 threads x 400 iterations per launch, {ROUNDS} launches per cell;")
print("# neighbour: 810 stem-shaped workgroups (512 threads, 150 KB LDS) relaunched before every victim launch; cell = wrong / checks")
print(f"{'pattern':72s}" + "".join(f"{n:>26s}" for n, _ in NEIGHBOURS))
for p, label in PATTERNS.items():
    row = f"{p:2d} {label:69s}"
    for name, mode in NEIGHBOURS:
        err = torch.zeros(32, dtype=torch.int32, device=dev)
        chk = torch.zeros(16, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        for r in range(ROUNDS):
            if mode == "chain16":
                with torch.cuda.stream(s_hog):
                    _chain(_cx)
            elif mode is not None:
                rc = hog.hog_launch(810, 153600, mode, 2000 if mode == 2 else 4000, sink.data_ptr(), s_hog.cuda_stream)
                assert rc == 0, rc
            rc = haz.hazard_launch(2048, p, 400, err.data_ptr(), chk.data_ptr(), s_vic.cuda_stream)
            assert rc == 0, rc
        torch.cuda.synchronize()
        total = 2048 * 256 * 400 * PER_ITER[p] * ROUNDS   # (the kernel's own 32-bit check counter wraps at this size)
        row += f"{int(err[p].item()) & 0xffffffff:>12d} /{total:>12.3g}"
        if p == 15 and int(err[15].item()):
            row += f" (lanes 0-15 / 16-31 / 32-47 / 48-63: {[int(v) for v in err[16:20].tolist()]})"
    print(row, flush=True)
