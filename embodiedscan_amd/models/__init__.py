"""Registry-visible model classes of the mv-3ddet hot path (mirrors embodiedscan.models)."""
from .backbones.mink_resnet import MinkResNet
from .backbones.resnet2d import ResNet
from .data_preprocessors.data_preprocessor import Det3DDataPreprocessor
from .dense_heads.fcaf3d_head import FCAF3DHeadRotMat
from .dense_heads.imvoxel_occ_head import ImVoxelOccHead
from .dense_heads.grounding_head import GroundingHead
from .detectors.dense_fusion_occ import DenseFusionOccPredictor
from .detectors.sparse_featfusion_grounder import SparseFeatureFusion3DGrounder
from .necks.mink_neck import MinkNeck
from .necks.fpn import FPN
from .necks.imvoxel_neck import IndoorImVoxelNeck
from .task_modules.anchor_3d_generator import AlignedAnchor3DRangeGenerator
from .detectors.sparse_featfusion_single_stage import SparseFeatureFusionSingleStage3DDetector

__all__ = ['MinkResNet', 'ResNet', 'Det3DDataPreprocessor', 'FCAF3DHeadRotMat',
           'SparseFeatureFusionSingleStage3DDetector', 'ImVoxelOccHead', 'DenseFusionOccPredictor', 'FPN', 'IndoorImVoxelNeck',
           'AlignedAnchor3DRangeGenerator', 'GroundingHead', 'SparseFeatureFusion3DGrounder', 'MinkNeck']
