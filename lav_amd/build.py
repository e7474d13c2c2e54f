"""Build liblav_amd.so (HIP kernels + C ABI) in-tree for gfx950.

    python -m lav_amd.build            # (re)build if sources are newer than the library
    python -m lav_amd.build --force

hipcc cross-compiles without a GPU; the .so is git-ignored but travels with the tree.
"""
from __future__ import annotations

import os
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "liblav_amd.so")
SOURCES = ["misc.hip", "pillar.hip", "paint.hip", "gru.hip", "gru_seq.hip", "conv.hip", "conv_f16.hip", "conv_pair.hip", "deconv.hip", "crop.hip", "frame.hip", "attn.hip", "bn_train.hip", "conv_wgrad.hip", "upconv.hip"]
# Parity-critical float32 arithmetic (cell ids, decoration, camera projection) must round every multiply and
# add separately, like the oracle: these translation units are compiled with FMA contraction off (the header
# helpers __fadd_rn/__fmul_rn are plain operators that clang would otherwise fuse after inlining).
EXTRA_FLAGS = {"pillar.hip": ["-ffp-contract=off"], "paint.hip": ["-ffp-contract=off"], "crop.hip": ["-ffp-contract=off"],
               "frame.hip": ["-ffp-contract=off"]}
# -fno-slp-vectorize (round 5): the SLP vectoriser turns pairs of scalar float operations into packed fp32 instructions
# (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32) and selects halves of register pairs with op_sel.  On gfx950 a packed fp32 instruction
# whose op_sel bit is set (low result from the HIGH register of a source pair) returns wrong values in lanes 48-63 when waves of a
# bf16-matrix + LDS heavy kernel share its SIMD (tools/lds_hazard.py pattern 55: 535 936 wrong of 1.26e10 beside ERFNet's 16-channel
# run, 0 alone) - the "finite but wrong" results of DESIGN 4.4c.  Without the vectoriser hipcc emits no packed fp32 at all; the
# kernels that use it on purpose (deconv.hip) write the instruction themselves with op_sel = 0.  tests/test_capi_host.py disassembles
# the library and fails on any packed fp32 instruction with an op_sel bit set.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-fno-slp-vectorize",
         "-I", os.path.join(REPO, "include"), "-I", CSRC]


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(REPO, "include", "lav_amd.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not needs_build():
        return LIB
    objs = []
    procs = []
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    t0 = time.time()
    for s in SOURCES:
        o = os.path.join(HERE, "build", s.replace(".hip", ".o"))
        objs.append(o)
        cmd = [_hipcc(), *FLAGS, *EXTRA_FLAGS.get(s, []), "-c", os.path.join(CSRC, s), "-o", o]
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {s}:\n{out}")
        if verbose and out.strip():
            print(out)
    link = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB]
    r = subprocess.run(link, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stdout)
    if verbose:
        print(f"built {LIB} in {time.time() - t0:.1f}s")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
