"""mmdet.FPN on the es_hip convolution engine (image neck of the occupancy detector:
configs/occupancy/mv-occ_8xb1_embodiedscan-occ-80class.py:35-38, called at
embodiedscan/models/detectors/dense_fusion_occ.py:147-154).  Channels-last row matrices: a lateral 1x1 conv is a row GEMM
with bias, the top-down pathway is one in-place nearest-upsample-add kernel per level, the output 3x3 convs run through
the static image-grid maps of the 2-D backbone.  mmdet is an un-vendored dependency: semantics restated (laterals ->
top-down F.interpolate(size=..., mode='nearest') adds -> 3x3 output convs, no norm / activation, num_outs == num_ins)."""
import os

from ... import engine as E
from ...registry import MODELS
from ..backbones.resnet2d import _Grid


FPN_DENSE = [os.environ.get('ES_FPN_DENSE', '1') != '0']      # A/B switch: the 3x3 output convolutions on the dense engine (flat grid)


@MODELS.register_module(name='mmdet.FPN')
class FPN:
    def __init__(self, in_channels, out_channels, num_outs, start_level=0, end_level=-1, add_extra_convs=False,
                 relu_before_extra_convs=False, no_norm_on_lateral=False, conv_cfg=None, norm_cfg=None, act_cfg=None,
                 upsample_cfg=None, init_cfg=None):
        assert start_level == 0 and end_level in (-1, len(in_channels) - 1) and not add_extra_convs
        assert num_outs == len(in_channels), 'extra max-pool levels are not used by the shipped configs'
        assert norm_cfg is None and act_cfg is None
        assert upsample_cfg is None or upsample_cfg.get('mode', 'nearest') == 'nearest'
        self.in_channels, self.out_channels, self.num_outs = list(in_channels), out_channels, num_outs
        self.grids = {}

    def bind(self, arena, prefix='neck.'):
        par = lambda n: E.Param(arena.p[prefix + n], arena.g.get(prefix + n))
        self.lat = [(par(f'lateral_convs.{i}.conv.weight'), par(f'lateral_convs.{i}.conv.bias'))
                    for i in range(len(self.in_channels))]
        self.out = [(par(f'fpn_convs.{i}.conv.weight'), par(f'fpn_convs.{i}.conv.bias'))
                    for i in range(len(self.in_channels))]
        return self

    def forward(self, feats, n_img, levels=None):
        """feats: [(Var (n_img*h*w, C_i), h, w)] from the backbone.  Returns [(Var (n_img*h*w, out_channels), h, w)] for the
        requested output `levels` (default: all).  The occupancy detector consumes level 0 only
        (dense_fusion_occ.py:149), so it asks for [0] and the unused 3x3 convs are not run."""
        L = len(feats)
        levels = list(range(L)) if levels is None else list(levels)
        lats = []
        for (x, h, w), (wt, b) in zip(feats, self.lat):
            lats.append(E.conv(x, wt, None, None, x.d.shape[0], bias=b))
        for i in range(L - 1, 0, -1):
            E.upsample_add_(lats[i - 1], lats[i], n_img, feats[i - 1][1], feats[i - 1][2], feats[i][1], feats[i][2])
        outs = []
        for i in levels:
            h, w = feats[i][1], feats[i][2]
            key = (n_img, h, w)
            if key not in self.grids:
                self.grids[key] = _Grid(n_img, h, w, lats[i].d.device)
            wt, b = self.out[i]
            # 3x3 / pad 1 on a dense image grid: the dense engine by address arithmetic (flat grid: Z = 0) where it takes the shape
            # (bf16 mode, 256 channels), the 9-wide image map elsewhere -- built only if a launch needs it
            dense = (n_img, h, w, 0, 3, 1, 1) if FPN_DENSE[0] else None
            outs.append((E.conv(lats[i], wt, None, None, n_img * h * w, bias=b, dense=dense,
                                maps=lambda g=self.grids[key]: g.conv_map(3, 1, 1)[:2]), h, w))
        return outs

    __call__ = forward
