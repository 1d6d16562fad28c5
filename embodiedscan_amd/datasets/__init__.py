from .embodiedscan_dataset import EmbodiedScanDataset
from .loader import ScanLoader, shard_indices
from .loading import ScanPipeline
from .mv_3dvg_dataset import MultiView3DGroundingDataset

__all__ = ['EmbodiedScanDataset', 'MultiView3DGroundingDataset', 'ScanLoader', 'ScanPipeline', 'shard_indices']
