"""AlignedAnchor3DRangeGenerator (embodiedscan/models/task_modules/anchor/anchor_3d_generator.py:246-354): the prior
voxel-centre grid of the occupancy detector.  Host-side (the grid is static: computed once, uploaded once), f32 torch
arithmetic in the reference's order (linspace over n+1 edges, + half a cell, first n) so the centres are the same floats."""
import torch
from ...registry import TASK_UTILS


@TASK_UTILS.register_module()
class AlignedAnchor3DRangeGenerator:
    def __init__(self, ranges, sizes=None, scales=None, rotations=None, custom_values=(), reshape_out=True,
                 size_per_range=True, align_corner=False):
        sizes = sizes if sizes is not None else [[3.9, 1.6, 1.56]]
        self.ranges = [list(r) for r in ranges]
        if size_per_range and len(sizes) != len(self.ranges):
            assert len(self.ranges) == 1
            self.ranges = self.ranges * len(sizes)
        self.sizes, self.scales = sizes, (scales if scales is not None else [1])
        self.rotations = rotations if rotations is not None else [0, 1.5707963]
        self.custom_values, self.reshape_out, self.size_per_range = custom_values, reshape_out, size_per_range
        self.align_corner = align_corner

    @property
    def num_levels(self):
        return len(self.scales)

    def anchors_single_range(self, feature_size, anchor_range, scale, sizes, rotations, device='cpu'):
        """feature_size (D, H, W) = (z, y, x) cells -> (D, H, W, n_sizes, n_rot, 7) anchors (x, y, z, sx, sy, sz, rot)"""
        if len(feature_size) == 2:
            feature_size = [1, feature_size[0], feature_size[1]]
        r = torch.tensor(anchor_range, dtype=torch.float32)
        axes = []
        for lo, hi, n in ((r[0], r[3], feature_size[2]), (r[1], r[4], feature_size[1]), (r[2], r[5], feature_size[0])):
            c = torch.linspace(float(lo), float(hi), n + 1)
            if not self.align_corner:
                c = c + (c[1] - c[0]) / 2
            axes.append(c[:n])
        xc, yc, zc = axes
        sz = torch.tensor(sizes, dtype=torch.float32).reshape(-1, 3) * scale
        rot = torch.tensor(rotations, dtype=torch.float32)
        D, H, W, S, R = len(zc), len(yc), len(xc), sz.shape[0], rot.shape[0]
        out = torch.empty((D, H, W, S, R, 7), dtype=torch.float32)
        out[..., 0] = xc.view(1, 1, W, 1, 1)
        out[..., 1] = yc.view(1, H, 1, 1, 1)
        out[..., 2] = zc.view(D, 1, 1, 1, 1)
        out[..., 3:6] = sz.view(1, 1, 1, S, 1, 3)
        out[..., 6] = rot.view(1, 1, 1, 1, R)
        return out.to(device)

    def single_level_grid_anchors(self, featmap_size, scale, device='cpu'):
        if not self.size_per_range:
            return self.anchors_single_range(featmap_size, self.ranges[0], scale, self.sizes, self.rotations, device)
        per = [self.anchors_single_range(featmap_size, r, scale, s, self.rotations, device)
               for r, s in zip(self.ranges, self.sizes)]
        return torch.cat(per, dim=-3)

    def grid_anchors(self, featmap_sizes, device='cpu'):
        assert self.num_levels == len(featmap_sizes)
        out = []
        for fs, sc in zip(featmap_sizes, self.scales):
            a = self.single_level_grid_anchors(fs, sc, device)
            out.append(a.reshape(-1, a.shape[-1]) if self.reshape_out else a)
        return out
