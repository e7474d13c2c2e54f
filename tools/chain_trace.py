"""In-kernel phase shares of ERFNet's persistent runs on fp16 pieces (LAV_PAIR_CHAIN_TRACE=1: shader-clock cycles per pair and workgroup, printed
by the library for every tenth launch of the 64- and 128-channel runs).

    LAV_PAIR_CHAIN_TRACE=1 python tools/chain_trace.py"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from lav_amd import _lib, ops, synth
from lav_amd.rgb import RGBSegmentationModel
DEV = torch.device("cuda", 0)
seg = RGBSegmentationModel([4, 6, 7, 10]); seg.load_state_dict(synth.seeded_state_dict(seg, prefix="seg.")); seg.eval().to(DEV)
cams, _ = synth.rgb_frames()
x = torch.tensor(np.stack([c[..., :3][..., ::-1] for c in cams], 0).copy()).permute(0, 3, 1, 2).float().to(DEV)
with torch.no_grad(), ops.precision(_lib.CONV_F16X3):
    for _ in range(12):
        seg(x)
    torch.cuda.synchronize()
