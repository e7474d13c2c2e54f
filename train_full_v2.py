#!/usr/bin/env python3
"""train_full_v2 (lav/train_full_v2.py): LiDARModel + UniPlanner end-to-end training (detection, segmentation and
distilled motion losses), one process per GPU.

    python train_full_v2.py --synthetic --batch-size 32 --steps 20
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 train_full_v2.py --synthetic ...

--batch-size is the GLOBAL batch (32 in BASELINE.json); each rank takes batch/world samples."""
from lav_amd.train.run import main

if __name__ == "__main__":
    main("lidar")
