"""Training path of the hot path's models: losses, the LAV trainer with its two steps
(train_bev: privileged BEVPlanner; train_lidar: LiDARModel + UniPlanner distilled from it), synthetic batches, and
data-parallel execution with one process per GPU over torch.distributed (RCCL on MI355X, gloo in CPU tests).

Convolutions and Linear layers run on torch autograd (MIOpen / rocBLAS); liblav_amd supplies the pillar front end
(lav_pillar_decorate), scatter_max with its backward - the torch_scatter replacement -, the rotated crops with their gather
backward, the GRU recurrences (forward + backward), BatchNorm on batch statistics fused with its ReLU / residual (hipnn.py),
the frozen teacher's inference kernels and the device-side re-packing behind the per-step log inference."""
from .losses import DetLoss, build_seg_mask, bev_losses, lidar_losses  # noqa: F401
from .lav import LAV, TrainConfig  # noqa: F401
from .synthetic import synthetic_bev_batch, synthetic_lidar_batch  # noqa: F401
