"""Data side of the trainers: recorded routes (LMDB) -> the batches `LAV.train_bev` / `LAV.train_lidar` take
(lav/utils/datasets of the reference)."""
from .datasets import get_data_loader  # noqa: F401
