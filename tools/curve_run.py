import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
from lav_amd.train import LAV, TrainConfig, synthetic_lidar_batch
g = np.load('/root/repo/tests/golden/train_curve.npz')
keys = [str(k) for k in g['keys']]
for rep in range(2):
    lav = LAV(TrainConfig(log_inference=False), torch.device('cuda'), what="lidar")
    batches = [synthetic_lidar_batch(2, seed=40 + i, max_points=20000, num_objs=3) for i in range(4)]
    rows = []
    for step in range(500):
        torch.manual_seed(1000 + step)
        info = lav.train_lidar(*batches[step % 4])
        rows.append([info[k] for k in keys])
    np.save(f'/root/repo/gpurun_out/curve_ours_{rep}.npy', np.array(rows))
