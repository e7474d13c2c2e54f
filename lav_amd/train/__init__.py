"""Training path of the hot path's models: losses, the LAV trainer with its two steps
(train_bev: privileged BEVPlanner; train_lidar: LiDARModel + UniPlanner distilled from it), synthetic batches, and
data-parallel execution with one process per GPU over torch.distributed (RCCL on MI355X, gloo in CPU tests).

Forward/backward of the dense layers runs on torch autograd (MIOpen / rocBLAS) in this round; liblav_amd supplies the
pillar front end (lav_pillar_decorate) and scatter_max with its backward - the torch_scatter replacement - and the
frozen teacher's inference kernels."""
from .losses import DetLoss, build_seg_mask, bev_losses, lidar_losses  # noqa: F401
from .lav import LAV, TrainConfig  # noqa: F401
from .synthetic import synthetic_bev_batch, synthetic_lidar_batch  # noqa: F401
