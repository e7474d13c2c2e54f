"""lav_amd - MI355X (gfx950) implementation of LAV's per-frame perception -> prediction -> planning
forward path behind the reference's team_code_v2 Python surface.

    from lav_amd import LiDARModel, UniPlanner, BEVPlanner, InferModel, PointPillarNet, CoordConverter

Hot kernels live in lav_amd/csrc (HIP) behind the C ABI of include/lav_amd.h (liblav_amd.so, built by
`python -m lav_amd.build`); this package is the host-side mirror of the reference's module interface.
"""
from .point_pillar import DynamicPointNet, PointPillarNet  # noqa: F401
from .lidar import ConvBackbone, Head, LiDARModel  # noqa: F401
from .resnet import ResNet, resnet18  # noqa: F401
from .bev_planner import BEVPlanner  # noqa: F401
from .uniplanner import UniPlanner  # noqa: F401
from .model_inference import CoordConverter, InferModel, crop_feature, extract_peak, transform_points  # noqa: F401

__all__ = ["DynamicPointNet", "PointPillarNet", "ConvBackbone", "Head", "LiDARModel", "ResNet", "resnet18",
           "BEVPlanner", "UniPlanner", "CoordConverter", "InferModel", "crop_feature", "extract_peak",
           "transform_points"]
