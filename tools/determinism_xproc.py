#!/usr/bin/env python3
"""Is the train_lidar step reproducible from PROCESS to process (tests/_curve_worker.py's conditions: deterministic switches, an empty
MIOpen user database)?  `dump` runs a few steps and stores the loss terms of every step and a checksum of every parameter gradient of
step 0; `cmp` names the first difference of two dumps.

    python tools/determinism_xproc.py dump a.npz [steps] ; python tools/determinism_xproc.py dump b.npz ; python tools/determinism_xproc.py cmp a.npz b.npz
"""
import os
import sys
import tempfile

import numpy as np

if sys.argv[1] == "cmp":
    a, b = np.load(sys.argv[2], allow_pickle=True), np.load(sys.argv[3], allow_pickle=True)
    ra, rb = a["rows"], b["rows"]
    first = next((i for i in range(len(ra)) if not np.array_equal(ra[i], rb[i])), None)
    print("first step whose loss terms differ:", first, "| terms:", list(a["keys"]))
    for i in range(len(ra)):
        d = np.abs(ra[i] - rb[i]) / np.maximum(np.abs(ra[i]), 1e-9)
        print(f"  step {i:2d} rel diff per term:", " ".join(f"{x:.1e}" for x in d))
    na, ga, gb = list(a["names"]), a["grads"], b["grads"]
    bad = [(n, x, y) for n, x, y in zip(na, ga, gb) if x != y]
    print(f"parameter gradients after step 0: {len(bad)} of {len(na)} checksums differ")
    groups = {}
    for n, x, y in bad:
        groups.setdefault(".".join(n.split(".")[:4]), []).append(abs(x - y) / max(abs(x), 1e-12))
    for k, v in sorted(groups.items()):
        print(f"  {k:60s} {len(v):3d} tensors, max rel diff {max(v):.2e}")
    sys.exit(0)

_db = tempfile.mkdtemp(prefix="lav_xproc_miopen_")
os.environ["MIOPEN_USER_DB_PATH"] = _db
os.environ["MIOPEN_CUSTOM_CACHE_DIR"] = _db
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from lav_amd.train import LAV, TrainConfig  # noqa: E402
from lav_amd.train.run import set_deterministic  # noqa: E402
from lav_amd.train.synthetic import synthetic_lidar_batch  # noqa: E402

out = sys.argv[2]
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 8
set_deterministic(True)
torch.manual_seed(0)
dev = torch.device("cuda")
lav = LAV(TrainConfig(log_inference=False), dev, what="lidar")
batches = [synthetic_lidar_batch(2, seed=40 + i, max_points=20000, num_objs=3) for i in range(4)]
rows, keys, names, grads = [], None, [], []
for s in range(steps):
    torch.manual_seed(1000 + s)
    info = lav.train_lidar(*batches[s % 4])
    keys = keys or [k for k, v in info.items() if isinstance(v, float)]
    rows.append([info[k] for k in keys])
    if s == 0:
        for n, p in lav.student.named_parameters():
            if p.grad is not None:
                names.append(n); grads.append(p.grad.detach().double().sum().item())
np.savez(out, rows=np.array(rows), keys=np.array(keys), names=np.array(names), grads=np.array(grads))
