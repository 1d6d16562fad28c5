from .anchor_3d_generator import AlignedAnchor3DRangeGenerator  # noqa: F401
