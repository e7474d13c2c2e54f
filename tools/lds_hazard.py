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
    16: "NO LDS: fused multiply-adds only, twice",
    17: "NO LDS: compares + selects (v_cmp -> mask -> v_cndmask), twice",
    18: "NO LDS: transcendental unit (v_exp / v_rcp / v_sin / v_sqrt), twice",
    19: "NO LDS: integer add / mul / shift / min / max, twice",
    20: "NO LDS: floor + float <-> int conversions, twice",
    21: "NO LDS: IEEE division (v_div_scale / v_div_fmas / v_div_fixup), twice",
    22: "NO LDS: v_cmp -> vcc -> v_cndmask back to back (asm), twice",
    23: "NO LDS: divergent branches (saveexec / restore around asm), twice",
    24: "NO LDS: v_cmp, v_cmp -> s_and_b64 -> v_cndmask (asm), twice",
    25: "NO LDS: precise sinf + cosf, twice",
    26: "NO LDS: precise expf + division, twice",
    27: "NO LDS: v_pk_fma_f32 chain (packed fp32), twice",
    28: "NO LDS: v_pk_mul_f32 + v_pk_add_f32 chain (packed fp32), twice",
    29: "NO LDS: the arithmetic of 27 on scalar v_fma_f32, twice",
    30: "NO LDS: v_cmp -> s[2:3] -> v_cndmask s[2:3] back to back (asm), twice",
    31: "NO LDS: two lane masks in SGPR pairs alive at once (asm), twice",
    42: "NO LDS: packed fp32 with an SGPR-pair source (v_pk_mul_f32 v, v, s[2:3]; v_pk_add_f32 v, v, s[4:5]), twice",
    43: "NO LDS: v_mul_f32 / v_add_f32 with a 32-bit SGPR source, twice",
    44: "NO LDS: v_pk_fma_f32 v, v, s[2:3], v, twice",
    45: "15 with every intermediate kept: which one differs first",
    46: "NO LDS: v_pk_add_f32 v, v, s[2:3] ; s_mov_b32 s2 ; s_mov_b32 s3 (SGPR write-after-read), twice",
    47: "NO LDS: 46 with two vector instructions before the s_mov (pattern 15's distance), twice",
    48: "NO LDS: v_add_f32 v, s2, v ; s_mov_b32 s2 (32-bit SGPR write-after-read), twice",
    49: "NO LDS: 46 with s_nop 4 between the read and the overwrite, twice",
    50: "NO LDS: pattern 15's instructions around gx as asm (cndmask x2, 4 packed ops with op_sel / neg, SGPR add), twice",
    51: "NO LDS: 50 with s_nop 3 between the instructions, twice",
    52: "NO LDS: 50 without the v_cndmask producers, twice",
    53: "NO LDS: 50 without the SGPR add, twice",
    54: "NO LDS: v_pk_mul_f32 op_sel_hi:[0,1] (+ plain v_pk_add_f32), twice",
    55: "NO LDS: v_pk_mul_f32 op_sel:[0,1] op_sel_hi:[0,0] (+ plain v_pk_add_f32), twice",
    56: "NO LDS: v_pk_fma_f32 neg_lo:[0,0,1] neg_hi:[0,0,1], twice",
    57: "NO LDS: v_pk_fma_f32 op_sel_hi:[0,1,1], twice",
    58: "NO LDS: plain v_pk_mul_f32 + v_pk_add_f32 op_sel_hi:[1,0], twice",
    59: "NO LDS: v_pk_fma_f32 without modifiers (control), twice",
    60: "NO LDS: v_pk_add_f32 op_sel:[0,1] op_sel_hi:[1,0] (second source swapped), twice",
    61: "NO LDS: v_pk_fma_f32 op_sel:[0,1,0] op_sel_hi:[1,0,1] (second source swapped), twice",
    62: "NO LDS: v_pk_mul_f32 op_sel:[1,0] op_sel_hi:[0,1] (first source swapped), twice",
    63: "NO LDS: v_pk_mul_f32 op_sel:[0,1] op_sel_hi:[1,1] (high register for both results), twice",
    40: "registers at REST: 24 VGPRs written once, the wave sleeps, re-reads them",
    41: "registers at rest while the wave runs an fma chain on other registers",
    32: "15 without sinf / cosf", 33: "15 without expf", 34: "15 without the masked weights (bool masks, s_and_b64, v_cndmask)",
    35: "15 without the integer clamps + int -> float", 36: "15 without the divisions", 37: "15 without sinf / cosf / expf",
    38: "15 without sinf / cosf / expf / divisions", 39: "15 without masked weights and integer clamps",
}
if os.environ.get("HAZARD_PATTERNS"):
    PATTERNS = {k: v for k, v in PATTERNS.items() if str(k) in os.environ["HAZARD_PATTERNS"].split(",")}
PER_ITER = {13: 6, 14: 6, 15: 1, 16: 1, 17: 1, 18: 1, 19: 1, 20: 1, 21: 1, 22: 1, 23: 1, 24: 1, 25: 1, 26: 1, 27: 1, 28: 1, 29: 1, 30: 1, 31: 1, 45: 1, 46: 1, 47: 1, 48: 1, 49: 1, 50: 1, 51: 1, 52: 1, 53: 1, 54: 1, 55: 1, 56: 1, 57: 1, 58: 1, 59: 1, 60: 1, 61: 1, 62: 1, 63: 1, 40: 24, 41: 24, 42: 1, 43: 1, 44: 1, 32: 1, 33: 1, 34: 1, 35: 1, 36: 1, 37: 1, 38: 1, 39: 1, 0: 1, 1: 1, 2: 2, 3: 1, 4: 2, 5: 8, 6: 2, 7: 1, 8: 2, 9: 2, 10: 4, 11: 4, 12: 2}
NEIGHBOURS = [("alone", None), ("matrix only (hog 0)", 0), ("matrix + LDS (hog 1)", 1), ("LDS only (hog 3)", 3), ("ERFNet 16-ch pair run", "chain16")]
# the frame's own strongest neighbour: ERFNet's 16-channel persistent pair run (432 row workgroups x 256 threads, 66-80 KB of LDS each:
# the only matrix + LDS kernel of the frame that leaves tens of KB of LDS free on its CUs) - tools/crop_victim.py
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch.nn as nn  # noqa: E402
from lav_amd.ops import Conv1dPair, Conv1dPairChain  # noqa: E402
_pairs = [Conv1dPair(nn.Conv2d(16, 16, (3, 1), padding=(1, 0)), nn.Conv2d(16, 16, (1, 3), padding=(0, 1)), nn.BatchNorm2d(16, eps=1e-3).eval(), device="cuda") for _ in range(10)]
_chain = Conv1dPairChain(_pairs, [i % 2 == 1 for i in range(10)])
_cx = torch.randn((3, 16, 144, 128), device="cuda")
from lav_amd import _lib as _lavlib  # noqa: E402
from lav_amd.ops import ConvLayer  # noqa: E402
_stem = ConvLayer(torch.randn(64, 384, 7, 7) / (384 * 49) ** 0.5, stride=2, padding=(3, 3), relu_post=True, precision=_lavlib.CONV_BF16X6, device="cuda")
_stem_x = torch.randn(15, 384, 96, 96, device="cuda")
_c64 = Conv1dPairChain([Conv1dPair(nn.Conv2d(64, 64, (3, 1), padding=(1, 0)), nn.Conv2d(64, 64, (1, 3), padding=(0, 1)), nn.BatchNorm2d(64, eps=1e-3).eval(), device="cuda") for _ in range(10)], [i % 2 == 1 for i in range(10)])
_c64x = torch.randn((3, 64, 72, 64), device="cuda")
NEIGHBOURS += [("7x7 stem (split kernel)", "stem"), ("ERFNet 64-ch pair run", "chain64")]
if os.environ.get("HAZARD_NEIGHBOURS"):
    NEIGHBOURS = [nb for nb in NEIGHBOURS if ("alone" if nb[1] is None else str(nb[1])) in os.environ["HAZARD_NEIGHBOURS"].split(",")]
s_hog, s_vic = torch.cuda.Stream(), torch.cuda.Stream()
sink = torch.zeros(16, device=dev)
print(f"# {torch.cuda.get_device_name(0)}; victim: 2048 workgroups x 256 threads x 400 iterations per launch, {ROUNDS} launches per cell;")
print("# neighbour: 810 stem-shaped workgroups (512 threads, 150 KB LDS) relaunched before every victim launch; cell = wrong / checks")
print(f"{'pattern':72s}" + "".join(f"{n:>26s}" for n, _ in NEIGHBOURS))
for p, label in PATTERNS.items():
    row = f"{p:2d} {label:69s}"
    for name, mode in NEIGHBOURS:
        err = torch.zeros(96, dtype=torch.int32, device=dev)
        chk = torch.zeros(16, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        for r in range(ROUNDS):
            if mode == "chain16":
                with torch.cuda.stream(s_hog):
                    _chain(_cx)
            elif mode == "stem":
                with torch.cuda.stream(s_hog):
                    _stem(_stem_x)
            elif mode == "chain64":
                with torch.cuda.stream(s_hog):
                    _c64(_c64x)
            elif mode is not None:
                rc = hog.hog_launch(810, 153600, mode, 2000 if mode == 2 else 4000, sink.data_ptr(), s_hog.cuda_stream)
                assert rc == 0, rc
            rc = haz.hazard_launch(2048, p, 400, err.data_ptr(), chk.data_ptr(), s_vic.cuda_stream)
            assert rc == 0, rc
        torch.cuda.synchronize()
        total = 2048 * 256 * 400 * PER_ITER[p] * ROUNDS   # (the kernel's own 32-bit check counter wraps at this size)
        row += f"{int(err[p].item()) & 0xffffffff:>12d} /{total:>12.3g}"
        if p >= 15 and int(err[p].item()):
            row += f" (lanes 0-15 / 16-31 / 32-47 / 48-63: {[int(v) for v in err[48:52].tolist()]})"
            if p == 45:
                row += f" first differing intermediate (cs sn k gx gy ix iy fx fy x0 y0 wx1 wy1 w00 w01 w10 w11 cx0 cy1 exp): {[int(v) for v in err[64:84].tolist()]}"
            if p in (40, 41):
                row += f" registers j & 7 hit: {[int(v) for v in err[56:64].tolist()]}"
    print(row, flush=True)
