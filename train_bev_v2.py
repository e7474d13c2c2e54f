#!/usr/bin/env python3
"""train_bev_v2 (lav/train_bev_v2.py): privileged BEVPlanner training, one process per GPU.

    python train_bev_v2.py --synthetic --batch-size 64 --steps 50
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 train_bev_v2.py --synthetic ...

Only synthetic batches are wired up (the LMDB readers are next); --batch-size is the GLOBAL batch."""
from lav_amd.train.run import main

if __name__ == "__main__":
    main("bev")
