"""Registry-visible model classes of the mv-3ddet hot path (mirrors embodiedscan.models)."""
from .backbones.mink_resnet import MinkResNet
from .backbones.resnet2d import ResNet
from .data_preprocessors.data_preprocessor import Det3DDataPreprocessor
from .dense_heads.fcaf3d_head import FCAF3DHeadRotMat
from .detectors.sparse_featfusion_single_stage import SparseFeatureFusionSingleStage3DDetector

__all__ = ['MinkResNet', 'ResNet', 'Det3DDataPreprocessor', 'FCAF3DHeadRotMat',
           'SparseFeatureFusionSingleStage3DDetector']
