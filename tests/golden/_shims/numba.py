"""lav/utils/datasets/temporal_lidar_painted_dataset.py:3 imports numba and never uses it."""
