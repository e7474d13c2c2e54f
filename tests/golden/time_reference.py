"""Times the REFERENCE's own per-frame GPU-side work on this container's CPU cores (informational):
seg model + softmax, InferModel.forward_paint, temporal stacking, InferModel.forward, brake model - the reference
modules imported from /root/reference with the same shims as make_golden.py, seeded weights, synthetic inputs.

    python tests/golden/time_reference.py        (needs /root/reference: container only, not the GPU box)
"""
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (sets up the paths and imports the reference modules)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from lav_amd import synth  # noqa: E402

torch.set_num_threads(os.cpu_count())
lm, up = mg.build_reference()
seg = mg.RGBSegmentationModel([4, 6, 7, 10]).eval(); seg.load_state_dict(synth.seeded_state_dict(seg, prefix="seg."))
bra = mg.RGBBrakePredictionModel([4, 6, 7, 10]).eval(); bra.load_state_dict(synth.seeded_state_dict(bra, prefix="bra."))
im = mg.ref_mi.InferModel(lm, up, 1.5, 2.4, device=torch.device("cpu"))
cams, tel = synth.rgb_frames()
rgbs = [c[..., :3][..., ::-1] for c in cams]
all_rgb = torch.tensor(np.stack(rgbs, 0).copy()).permute(0, 3, 1, 2).float()
wide = torch.tensor(np.concatenate(rgbs, axis=1)[None].copy()).permute(0, 3, 1, 2).float()
tel_rgb = torch.tensor(tel[..., :3][..., ::-1][:-96][None].copy()).permute(0, 3, 1, 2).float()
ticks = [torch.from_numpy(synth.lidar_sweep(32768, name=f"tick{i}")) for i in range(2)]
nxp = torch.tensor([0.0, -10.0])
stages = {}


def timed(name, fn):
    t0 = time.perf_counter()
    out = fn()
    stages.setdefault(name, []).append(time.perf_counter() - t0)
    return out


frames = []
with torch.no_grad():
    history = []
    for it in range(5):
        t_frame = time.perf_counter()
        cur = torch.cat(ticks)
        x, y, z = cur[:, 0], cur[:, 1], cur[:, 2]
        cur = cur[~((x > -2.4) & (x < 0) & (y > -0.8) & (y < 0.8) & (z > -1.5) & (z < -1))]
        sem = timed("seg+softmax", lambda: torch.softmax(seg(all_rgb), dim=1))
        fused = timed("paint", lambda: im.forward_paint(cur, sem))
        history = (history + [fused])[-3:]
        def stack():
            parts = []
            for i, l in enumerate(reversed(history)):
                oh = torch.zeros((len(l), 3)); oh[:, i] = 1
                parts.append(torch.cat([l, oh], dim=-1))
            while len(parts) < 3:
                parts.append(parts[-1])
            return torch.cat(parts)
        pts = timed("stack", stack)
        timed("InferModel.forward", lambda: im(pts, nxp, 3))
        timed("brake", lambda: bra(wide, tel_rgb))
        frames.append(time.perf_counter() - t_frame)
med = float(np.median(frames[1:]))
print(f"reference per-frame work on {os.cpu_count()} CPU threads: {med * 1e3:.0f} ms/frame = {1 / med:.2f} frames/s")
for k, v in stages.items():
    print(f"  {k:22s} {np.median(v[1:]) * 1e3:8.1f} ms")
