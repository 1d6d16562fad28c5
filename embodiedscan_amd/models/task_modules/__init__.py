from .anchor_3d_generator import AlignedAnchor3DRangeGenerator  # noqa: F401
from .assigners import BBox3DL1Cost, BinaryFocalLossCost, HungarianAssigner3D, IoU3DCost  # noqa: F401
