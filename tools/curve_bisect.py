"""Which piece of the MI355X training graph moves the 500-step loss curve of BASELINE config #5?  

Runs tests/test_gpu_train.py's curve (four alternating seeded batches of 2 x 20 000-point clouds, 500 steps, deterministic
algorithms) with ONE piece of the training graph swapped for torch's own ops and prints the same summary figures, so that
the variants line up in a table:

    python tools/curve_bisect.py base | bn | gru | crop | pillar | all | nowino | nondet [--steps 500] [--out file.npy]

base    the product: liblav_amd's autograd functions (BatchNorm+ReLU, sequence GRU, indexed rotated crops, pillar decorate /
        scatter-max), convolutions on MIOpen
bn      nn.BatchNorm2d + F.relu instead of lav_bn_train_*            (LAV_TRAIN_BN=torch)
gru     nn.GRU (MIOpen RNN) instead of lav_gru_seq_*                 (LAV_TRAIN_GRU=torch)
crop    affine_grid + grid_sample instead of lav_crop_rotate_indexed (LAV_TRAIN_CROP=torch)
pillar  torch index ops (unique / index_add / index_reduce) instead of lav_pillar_decorate + lav_scatter_max
all     all four
nowino  base with MIOpen's Winograd convolution solvers switched off (MIOPEN_DEBUG_CONV_WINOGRAD=0)
nondet  base without set_deterministic()
"""
import argparse
import os
import sys

ap = argparse.ArgumentParser()
ap.add_argument("variant")
ap.add_argument("--steps", type=int, default=500)
ap.add_argument("--out", default=None)
a = ap.parse_args()
v = a.variant
if v in ("bn", "all"):
    os.environ["LAV_TRAIN_BN"] = "torch"
if v in ("gru", "all"):
    os.environ["LAV_TRAIN_GRU"] = "torch"
if v in ("crop", "all"):
    os.environ["LAV_TRAIN_CROP"] = "torch"
if v == "nowino":
    os.environ["MIOPEN_DEBUG_CONV_WINOGRAD"] = "0"

import numpy as np  # noqa: E402
import torch  # noqa: E402

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from lav_amd.point_pillar import PointPillarNet  # noqa: E402
from lav_amd.train import LAV, TrainConfig, synthetic_lidar_batch  # noqa: E402
from lav_amd.train.run import set_deterministic  # noqa: E402


def pillar_forward_train_torch(self, lidar_list, num_points):
    """lav/models/point_pillar.py:92-116 with torch ops on the device (the torch_scatter calls as index_add / index_reduce)."""
    pts, n = self._pack(lidar_list, num_points)
    B = pts.shape[0]
    rows, coords = [], []
    for b in range(B):
        p = pts[b, : n[b]]
        x, y = p[:, 0], p[:, 1]
        keep = (x >= self.min_x) & (x < self.max_x) & (y >= self.min_y) & (y < self.max_y)
        p = p[keep]
        xi = ((p[:, 0] - self.min_x) * self.pixels_per_meter).long()
        yi = ((p[:, 1] - self.min_y) * self.pixels_per_meter).long()
        rows.append(p)
        coords.append(torch.stack([torch.full_like(xi, b), xi, yi], dim=1))
    p, c = torch.cat(rows), torch.cat(coords)
    with torch.no_grad():
        uniq, inv = c.unique(return_inverse=True, dim=0)
        P = uniq.shape[0]
        xyz = p[:, :3]
        s = torch.zeros((P, 3), dtype=xyz.dtype, device=xyz.device).index_add_(0, inv, xyz)
        cnt = torch.zeros((P,), dtype=xyz.dtype, device=xyz.device).index_add_(0, inv, torch.ones_like(xyz[:, 0]))
        mean = s / cnt.clamp(min=1)[:, None]
        xc = uniq[inv][:, 2].float() / self.pixels_per_meter + self.min_x
        yc = uniq[inv][:, 1].float() / self.pixels_per_meter + self.min_y
        dec = torch.cat([p, xyz - mean[inv], (xyz[:, 0] - xc)[:, None], (xyz[:, 1] - yc)[:, None]], dim=1)
    feat = self.point_net.net(dec)
    fmax = torch.full((P, feat.shape[1]), float("-inf"), dtype=feat.dtype, device=feat.device).index_reduce(0, inv, feat, "amax", include_self=True)
    fmax = torch.where(torch.isinf(fmax), torch.zeros_like(fmax), fmax)
    canvas = torch.zeros((B, feat.shape[1], self.ny, self.nx), dtype=feat.dtype, device=feat.device)
    canvas[uniq[:, 0], :, torch.clamp(self.ny - 1 - uniq[:, 1], 0, self.ny - 1), torch.clamp(uniq[:, 2], 0, self.nx - 1)] = fmax
    return canvas


if v in ("pillar", "all"):
    PointPillarNet.forward_train = pillar_forward_train_torch

g = np.load(os.path.join(REPO, "tests/golden/train_curve.npz"))
ref, keys = g["terms"][: a.steps], [str(k) for k in g["keys"]]
dev = torch.device("cuda")
if v != "nondet":
    set_deterministic(True)
torch.manual_seed(0)
lav = LAV(TrainConfig(log_inference=False), dev, what="lidar")
batches = [synthetic_lidar_batch(2, seed=40 + i, max_points=20000, num_objs=3) for i in range(4)]
rows = []
for step in range(a.steps):
    torch.manual_seed(1000 + step)
    info = lav.train_lidar(*batches[step % 4])
    rows.append([info[k] for k in keys])
ours = np.array(rows)
if a.out:
    np.save(a.out, ours)
smooth = lambda x, w: np.convolve(x, np.ones(w) / w, mode="valid")
to, tr = ours.sum(1), ref.sum(1)
blocks = " ".join(f"{to[i:i + 100].mean() / tr[i:i + 100].mean():.2f}" for i in range(0, a.steps - 99, 100))
whole = np.abs(smooth(to, 100) - smooth(tr, 100)) / smooth(tr, 100) if a.steps >= 200 else np.zeros(1)
plan0 = ours[0::4, keys.index("plan_loss")][-25:].mean()
print(f"{v:7s} step0 {to[0]:.3f} (ref {tr[0]:.3f})  block ratios {blocks}  max dev of 100-step MA {whole.max():.3f}  final {to[-100:].mean():.2f} vs {tr[-100:].mean():.2f}"
      f" = {to[-100:].mean() / tr[-100:].mean():.2f}x  batch-0 plan loss (last 25) {plan0:.2f} (ref {ref[0::4, keys.index('plan_loss')][-25:].mean():.2f})", flush=True)
