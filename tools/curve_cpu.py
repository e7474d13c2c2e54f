"""Config #5's 500-step curve on the HOST: the same trainer (lav_amd/train/lav.py) on torch CPU ops with oracle/train_cpu.py's
stand-ins for the two HIP-only pieces - a second sample of the reference's CPU run (tests/golden/train_curve.npz), to tell the
spread of a chaotic 500-step Adam trajectory apart from a bias of the MI355X kernels.

    python tools/curve_cpu.py --threads 8 --steps 500 --out gpurun_out/r4/curve_cpu_t8.npy [--batch 2 --points 20000]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

ap = argparse.ArgumentParser()
ap.add_argument("--threads", type=int, default=8)
ap.add_argument("--steps", type=int, default=500)
ap.add_argument("--batch", type=int, default=2)
ap.add_argument("--points", type=int, default=20000)
ap.add_argument("--objs", type=int, default=3)
ap.add_argument("--nbatches", type=int, default=4)
ap.add_argument("--out", required=True)
a = ap.parse_args()
torch.set_num_threads(a.threads)

from lav_amd.point_pillar import PointPillarNet  # noqa: E402
from lav_amd.train import LAV, TrainConfig, synthetic_lidar_batch  # noqa: E402
from oracle import train_cpu  # noqa: E402

PointPillarNet.forward_train = lambda self, lidars, num_points: train_cpu.pillar_forward_train(self, lidars, num_points)
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests/golden/train_curve.npz"))
keys = [str(k) for k in g["keys"]]
torch.manual_seed(0)
lav = LAV(TrainConfig(log_inference=False), torch.device("cpu"), what="lidar")
batches = [synthetic_lidar_batch(a.batch, seed=40 + i, max_points=a.points, num_objs=a.objs) for i in range(a.nbatches)]
rows = []
t0 = time.time()
with train_cpu.teacher_on_cpu(lav.bev_planner):
    for step in range(a.steps):
        torch.manual_seed(1000 + step)
        info = lav.train_lidar(*batches[step % a.nbatches])
        rows.append([info[k] for k in keys])
        if step % 20 == 0:
            np.save(a.out, np.array(rows))
            print(step, round(time.time() - t0), round(sum(rows[-1]), 3), flush=True)
np.save(a.out, np.array(rows))
