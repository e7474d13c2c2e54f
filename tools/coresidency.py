"""Victim x aggressor table for the co-residency effect of DESIGN 4.4c.

Every kernel of the frame that uses LDS ("victim") is launched N times on one stream while a second stream loops one of the
kernels known or suspected to disturb LDS-dependent results of waves sharing their CUs ("aggressor"); every launch is compared bit
for bit with the same launch on a quiet chip.  Two alternating inputs per victim, so that a value left over from the previous
launch would show.  One process = one setting of the aggressors' LDS claims (lav::lds_claim reads LAV_LDS_EXCLUSIVE once):

    python tools/coresidency.py [launches]                        # exact LDS sizes (the library's default since round 5)
    LAV_LDS_EXCLUSIVE=1 python tools/coresidency.py [launches]    # round 4's claims: the aggressors take their CU's LDS
    LAV_LDS_EXCLUSIVE=2 ...                                       # every claiming kernel takes the CU's whole LDS (also the two-per-CU ones)

Output: one line per (victim, aggressor): wrong / launches; profiles/r05_coresidency.md is assembled from the two runs.
"""
import ctypes
import os
import subprocess
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lav_amd import _lib, ops, synth  # noqa: E402
from lav_amd.ops import ConvLayer  # noqa: E402

dev = torch.device("cuda")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
ONLY_V = os.environ.get("CORES_VICTIMS", "").split(",") if os.environ.get("CORES_VICTIMS") else None
ONLY_A = os.environ.get("CORES_AGGRESSORS", "").split(",") if os.environ.get("CORES_AGGRESSORS") else None
torch.manual_seed(0)
g = lambda *s, sc=1.0: (torch.randn(*s) * sc).to(dev)

# ------------------------------------------------------------------------------------------------------------------ victims
victims = {}

H, T, NC = 512, 20, 6
_pw = (g(3 * H, 4, sc=0.3), g(3 * H, H, sc=H ** -0.5), g(3 * H, sc=0.1), g(3 * H, sc=0.1), g(2, H, sc=0.05), g(2, sc=0.1))
_pin = [(g(1, H, sc=0.5), g(1, 2, sc=3.0), g(1, NC, T, 2, sc=2.0)) for _ in range(2)]


def _plan(impl):
    def run(i):
        os.environ["LAV_PLAN_IMPL"] = impl
        e, n, c = _pin[i & 1]
        out = ops.gru_plan(e, n, c, *_pw, 5, 3, 4.0, 192.0)
        os.environ.pop("LAV_PLAN_IMPL")
        return out
    return run


victims["plan (k_plan_wave, no LDS)"] = _plan("wave")
victims["plan (k_plan_persistent, LDS: rounds 2-4)"] = _plan("lds")
victims["plan (k_plan_step x 100)"] = _plan("steps")

_cw = (g(NC, 192, 512, sc=512 ** -0.5), g(NC, 192, 64, sc=0.125), g(NC, 192, sc=0.1), g(NC, 192, sc=0.1), g(NC, 2, 64, sc=0.1), g(NC, 2, sc=0.1))
_cin = [g(7, 512, 3, 3).abs() for _ in range(2)]
_cmd = (g(NC, 512, sc=0.05), g(NC, sc=0.1))
_oris, _locs = g(7), g(7, 2, sc=5.0)
victims["embed_cast (k_gru_cast)"] = lambda i: torch.cat([t.flatten() for t in ops.embed_cast(_cin[i & 1], *_cw, T, cmd_w=_cmd[0], cmd_b=_cmd[1], oris=_oris, locs=_locs)])

from lav_amd.rgb import Attention  # noqa: E402
_att = Attention(512).eval().to(dev)
with torch.no_grad():
    for p_ in _att.parameters():
        p_.normal_(0, 0.05)
_ain = [g(1, 512, 9, 24), g(1, 512, 9, 24)]
victims["attn_pool (k_attn_pool)"] = lambda i: _att(_ain[i & 1])

_dconv = ConvLayer(torch.randn(128, 128, 3, 3) * 0.03, stride=1, padding=(1, 1), relu_post=True, precision=_lib.CONV_F32, device=dev)
_din = [g(7, 128, 12, 12), g(7, 128, 12, 12)]
victims["conv 128->128 @12x12 x7 (k_conv_direct)"] = lambda i: _dconv(_din[i & 1])

_tconv = ConvLayer(torch.randn(64, 64, 3, 3) * 0.04, stride=1, padding=(1, 1), relu_post=True, precision=_lib.CONV_F32, device=dev)
_tin = [g(1, 64, 160, 160), g(1, 64, 160, 160)]
victims["conv 64->64 @160x160 fp32 (tiled / direct plan)"] = lambda i: _tconv(_tin[i & 1])

_feat = [g(1, 384, 160, 160), g(1, 384, 160, 160)]
_clocs, _coris = (torch.rand(7, 2, device=dev) * 40 - 20), (torch.rand(7, device=dev) * 2 - 1)
victims["crop_rotate (k_crop_rotate_staged)"] = lambda i: ops.crop_rotate(_feat[i & 1], _clocs, _coris, 4.0, 96, 0.0, 0.75)

_heat = [torch.randn(2, 320, 320, device=dev) * 2 - 3 for _ in range(2)]
_size, _ori = g(2, 320, 320), g(2, 320, 320)
victims["extract_peaks (k_extract_peaks)"] = lambda i: ops.extract_peaks(_heat[i & 1], _size, _ori, apply_sigmoid=True)

from lav_amd.point_pillar import PointPillarNet  # noqa: E402
_ppn = PointPillarNet(16, (64, 64), -10, 70, -40, 40, 4).eval().to(dev)
_ppn.load_state_dict(synth.seeded_state_dict(_ppn))
_clouds = [torch.from_numpy(synth.stacked_lidar(65536, seed=s)).to(dev) for s in (11, 12)]
victims["pillar (k_bin + k_rows, 196 608 points)"] = lambda i: _ppn([_clouds[i & 1]], [_clouds[i & 1].shape[0]])

_sem = torch.from_numpy(synth.semantic_probs(seed=3)).to(dev)
_pl = [torch.from_numpy(synth.lidar_sweep(65536, seed=s)).to(dev) for s in (5, 6)]

# ------------------------------------------------------------------------------------------------------------------ aggressors
aggressors = {}
_stem = ConvLayer(torch.randn(64, 384, 7, 7) / (384 * 49) ** 0.5, stride=2, padding=(3, 3), relu_post=True, precision=_lib.CONV_BF16X6, device=dev)
_stem_x = torch.randn(15, 384, 96, 96, device=dev)
aggressors["stem 7x7 s2 x15 (tap-pair split kernel)"] = lambda: _stem(_stem_x)
_head = ConvLayer(torch.randn(256, 384, 3, 3) / (384 * 9) ** 0.5, stride=1, padding=(1, 1), relu_post=True, precision=_lib.CONV_BF16X6, device=dev)
_head_x = torch.randn(1, 384, 160, 160, device=dev)
aggressors["head conv 384->256 @160x160 (k_conv_split<2,2,2,2,2>)"] = lambda: _head(_head_x)
from lav_amd.rgb import RGBSegmentationModel  # noqa: E402
_seg = RGBSegmentationModel([4, 6, 7, 10]); _seg.load_state_dict(synth.seeded_state_dict(_seg, prefix="seg."))
_seg = _seg.eval().to(dev)
_seg_x = torch.rand(3, 3, 288, 256, device=dev) * 255
aggressors["ERFNet (persistent pair runs + its other layers)"] = lambda: _seg(_seg_x)

# ERFNet's persistent runs one width at a time (lav_conv1d_pair_chain: bf16 matrix instructions + LDS, row-resident workgroups that
# claim their CU's LDS - all of it at 64 / 128 channels, HALF of it at 16 channels, where 432 rows need two workgroups per CU)
import torch.nn as nn  # noqa: E402
from lav_amd.ops import Conv1dPair, Conv1dPairChain  # noqa: E402


def _chain(ch, h, w, dil):
    pairs = [Conv1dPair(nn.Conv2d(ch, ch, (3, 1), padding=(d, 0), dilation=(d, 1)), nn.Conv2d(ch, ch, (1, 3), padding=(0, d), dilation=(1, d)),
                        nn.BatchNorm2d(ch, eps=1e-3).eval(), device=dev) for d in dil]
    chain = Conv1dPairChain(pairs, [i % 2 == 1 for i in range(len(pairs))])
    x = torch.randn((3, ch, h, w), device=dev)
    assert chain.supported(x), (ch, h, w)
    return lambda: chain(x)


aggressors["pair chain 16 ch @144x128 x3 (432 rows, two per CU: half claims)"] = _chain(16, 144, 128, (1,) * 10)
aggressors["pair chain 64 ch @72x64 x3 (216 rows, one per CU: whole claim)"] = _chain(64, 72, 64, (1,) * 10)
aggressors["pair chain 128 ch @36x32 x3 (108 rows, one per CU: whole claim)"] = _chain(128, 36, 32, (1, 2, 1, 4, 1, 8, 1, 16))


def _erfnet_single_pairs():
    os.environ["LAV_ERFNET_CHAIN"] = "0"
    _seg(_seg_x)
    os.environ.pop("LAV_ERFNET_CHAIN")


aggressors["ERFNet, pairs launched one by one (LAV_ERFNET_CHAIN=0)"] = _erfnet_single_pairs

here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "probes")
so = os.path.join(here, "liblds_hog.so")
if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(os.path.join(here, "lds_hog.hip")):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", os.path.join(here, "lds_hog.hip"), "-o", so])
hoglib = ctypes.CDLL(so)
hoglib.hog_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
_sink = torch.zeros(16, device=dev)


def _synth():
    rc = hoglib.hog_launch(810, 153600, 1, 4000, _sink.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert rc == 0, rc


aggressors["lds_hog mode 1 (matrix + LDS, 150 KB: 10 KB left per CU)"] = _synth

# ------------------------------------------------------------------------------------------------------------------ run
s_hog, s_vic = torch.cuda.Stream(), torch.cuda.Stream()
print(f"# {torch.cuda.get_device_name(0)}  LAV_LDS_EXCLUSIVE={os.environ.get('LAV_LDS_EXCLUSIVE', '(default: 0 = exact LDS sizes)')}  launches per cell {N}", flush=True)
for vname, vic in victims.items():
    if ONLY_V and not any(k in vname for k in ONLY_V):
        continue
    with torch.no_grad(), torch.cuda.stream(s_vic):
        want = [vic(0).clone(), vic(1).clone()]
        again = [vic(0).clone(), vic(1).clone()]
    torch.cuda.synchronize()
    solo_ok = all(torch.equal(a, b) for a, b in zip(want, again))
    for aname, agg in [("(quiet chip)", None)] + list(aggressors.items()):
        if agg is not None and ONLY_A and not any(k in aname for k in ONLY_A):
            continue
        outs = []
        with torch.no_grad():
            if agg is not None:
                with torch.cuda.stream(s_hog):
                    agg()
            torch.cuda.synchronize()
            for i in range(N):
                if agg is not None:
                    with torch.cuda.stream(s_hog):
                        agg()
                with torch.cuda.stream(s_vic):
                    outs.append(vic(i).clone())
        torch.cuda.synchronize()
        bad = [(i, float((o - want[i & 1]).abs().max())) for i, o in enumerate(outs) if not torch.equal(o, want[i & 1])]
        worst = max([b[1] for b in bad], default=0.0)
        print(f"{vname:48s} | {aname:58s} | wrong {len(bad):4d} / {N}  max |diff| {worst:.3e}{'' if solo_ok else '  (NOT deterministic alone)'}", flush=True)
