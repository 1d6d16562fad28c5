from .embodiedscan_dataset import EmbodiedScanDataset
from .loader import ScanLoader, shard_indices
from .loading import ScanPipeline

__all__ = ['EmbodiedScanDataset', 'ScanLoader', 'ScanPipeline', 'shard_indices']
