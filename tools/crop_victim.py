"""What do the wrong values look like?  lav_crop_rotate (k_crop_rotate_staged: the touched box of the map staged through LDS eight
channels at a time, plain ds_write_b32 / ds_read_b32 and two workgroup barriers per sub-chunk) beside ERFNet's 16-channel persistent
pair run (432 row workgroups of 256 threads, bf16 matrix instructions + 16-byte LDS traffic, 66-80 KB of LDS each: the one neighbour
of the frame that leaves room for a 32 KB workgroup on its CU) - profiles/r05_coresidency.md.  For every wrong launch: how many
elements differ, where, and whether the wrong value is (a) the right value of the same pixel 8 channels earlier (= the previous
sub-chunk still in LDS: a stale read), (b) another pixel's value of the same channel, (c) zero, (d) something else.

    python tools/crop_victim.py [launches]        LAV_CROP_FWD_GENERAL=1: the gathering kernel (no LDS) as the control
"""
import os
import sys

import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lav_amd import ops  # noqa: E402
from lav_amd.ops import Conv1dPair, Conv1dPairChain  # noqa: E402

dev = torch.device("cuda")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 400
torch.manual_seed(0)
feat = torch.randn(1, 384, 160, 160, device=dev)
locs, oris = (torch.rand(7, 2, device=dev) * 40 - 20), (torch.rand(7, device=dev) * 2 - 1)
crop = lambda: ops.crop_rotate(feat, locs, oris, 4.0, 96, 0.0, 0.75)
ch, h, w = 16, 144, 128
pairs = [Conv1dPair(nn.Conv2d(ch, ch, (3, 1), padding=(1, 0)), nn.Conv2d(ch, ch, (1, 3), padding=(0, 1)), nn.BatchNorm2d(ch, eps=1e-3).eval(), device=dev) for _ in range(10)]
chain = Conv1dPairChain(pairs, [i % 2 == 1 for i in range(10)])
x = torch.randn((3, ch, h, w), device=dev)
s_hog, s_vic = torch.cuda.Stream(), torch.cuda.Stream()
with torch.cuda.stream(s_vic):
    want = crop().clone()
with torch.cuda.stream(s_hog):
    chain(x)
torch.cuda.synchronize()
outs = []
for i in range(N):
    with torch.cuda.stream(s_hog):
        chain(x)
    with torch.cuda.stream(s_vic):
        outs.append(crop().clone())
torch.cuda.synchronize()
bad = [i for i, o in enumerate(outs) if not torch.equal(o, want)]
print(f"{torch.cuda.get_device_name(0)}  LAV_CROP_FWD_GENERAL={os.environ.get('LAV_CROP_FWD_GENERAL', '0')}  LAV_LDS_EXCLUSIVE={os.environ.get('LAV_LDS_EXCLUSIVE', '1')}: "
      f"wrong launches {len(bad)} / {N}")
kinds = {"stale (same pixel, 8 channels earlier)": 0, "stale (same pixel, 8 channels later)": 0, "other pixel of the channel's crop": 0, "zero": 0, "other": 0}
nel = []
for i in bad[:40]:
    o = outs[i]
    d = (o != want).nonzero()
    nel.append(d.shape[0])
    tiles = set()
    for n_, c_, y_, x_ in d.tolist()[:4000]:
        got = float(o[n_, c_, y_, x_])
        tiles.add((n_, c_ // 32, c_ % 32 // 8, y_ // 16, x_ // 16))
        if c_ >= 8 and got == float(want[n_, c_ - 8, y_, x_]):
            kinds["stale (same pixel, 8 channels earlier)"] += 1
        elif c_ + 8 < want.shape[1] and got == float(want[n_, c_ + 8, y_, x_]):
            kinds["stale (same pixel, 8 channels later)"] += 1
        elif got == 0.0:
            kinds["zero"] += 1
        elif bool((want[n_, c_] == got).any()):
            kinds["other pixel of the channel's crop"] += 1
        else:
            kinds["other"] += 1
    if len(nel) <= 6:
        first = d[0].tolist()
        print(f"  launch {i}: {d.shape[0]} wrong elements in {len(tiles)} (crop, channel block, sub-chunk, tile y, tile x) units {sorted(tiles)[:6]}; "
              f"first at {first}: got {float(o[tuple(first)]):.6f} want {float(want[tuple(first)]):.6f}; "
              f"channels hit {sorted(set(d[:, 1].tolist()))[:12]}, rows {sorted(set(d[:, 2].tolist()))[:8]}, cols {sorted(set(d[:, 3].tolist()))[:8]}")
print("wrong elements per wrong launch (first 40):", nel)
print("classification of the wrong values:", kinds)
