"""ImVoxelOccHead (embodiedscan/models/dense_heads/imvoxel_occ_head.py:20-184) on the MI355X kernels: per level a 1x1x1
Conv3d (128 -> num_classes, no bias) as a row GEMM, the multi-scale supervision scatter (+ MaxPool3d of the visibility
mask) as two tiny kernels, and CrossEntropy(ignore 255) + sem_scal_loss + geo_scal_loss fused into one statistics pass,
one coefficient block and one gradient pass per level (es_occ_loss) that also seeds the gradient of the logits."""
import torch
from ... import engine as E
from ... import hip
from ...hip import P, call
from ...registry import MODELS


@MODELS.register_module()
class ImVoxelOccHead:
    def __init__(self, *args, num_classes=21, volume_h=40, volume_w=40, volume_z=16, in_channels=128, use_semantic=True,
                 train_cfg=None, test_cfg=None, **kwargs):
        assert use_semantic, 'the shipped occupancy config predicts semantic classes (use_semantic=True)'
        self.num_classes = num_classes
        self.volume_h, self.volume_w, self.volume_z = volume_h, volume_w, volume_z
        self.in_channels = list(in_channels) if isinstance(in_channels, (list, tuple)) else [in_channels]
        self.use_semantic = use_semantic
        self.training = True

    def bind(self, arena, prefix='bbox_head.'):
        self.occ = [E.Param(arena.p[f'{prefix}occ.{i}.weight'], arena.g.get(f'{prefix}occ.{i}.weight'))
                    for i in range(len(self.in_channels))]
        return self

    def forward(self, mlvl_feats, input_metas=None):
        """mlvl_feats: [(Var (n_vox_i, C_i), dims_i)] -> [(logits Var (n_vox_i, num_classes), dims_i)]"""
        return [(E.conv(f, w, None, None, f.d.shape[0]), dims) for (f, dims), w in zip(mlvl_feats, self.occ)]

    __call__ = forward

    def predict(self, x, batch_data_samples):
        """argmax over softmax of the finest level: (B, X, Y, Z) int64 (imvoxel_occ_head.py:93-108)"""
        prev = E.TAPE.enabled
        E.TAPE.enabled = False
        try:
            logits, dims = self.forward(x[:1])[0]
        finally:
            E.TAPE.enabled = prev
        n = logits.d.shape[0]
        out = torch.empty(n, dtype=torch.int32, device=logits.d.device)
        call('es_row_argmax', P(logits.d), logits.d.shape[1], n, self.num_classes, P(out), hip.stream())
        B = n // (dims[0] * dims[1] * dims[2])
        return out.view(B, *dims).long()

    def targets(self, gt_occupancy, gt_masks, ratio, dims, dev):
        """occ_multiscale_supervision (occ_loss.py:7-36) for every sample -> (B*X*Y*Z,) int32 on the device"""
        X, Y, Z = dims
        B = len(gt_occupancy)
        gt = torch.empty(B * X * Y * Z, dtype=torch.int32, device=dev)
        scratch = torch.empty(X * Y * Z, dtype=torch.int32, device=dev)
        for b in range(B):
            occ = gt_occupancy[b]
            occ = occ if (occ.is_cuda and occ.dtype == torch.int32) else occ.to(dev).to(torch.int32)
            occ = occ.contiguous()
            m = None
            if gt_masks is not None:
                m = gt_masks[b]
                m = (m if m.is_cuda else m.to(dev)).to(torch.uint8).contiguous()
                assert tuple(m.shape) == (X * ratio, Y * ratio, Z * ratio)
            call('es_occ_targets', P(occ), occ.shape[0], int(ratio), X, Y, Z, P(m), P(scratch),
                 gt.data_ptr() + 4 * b * X * Y * Z, hip.stream())
        return gt

    def loss(self, x, batch_data_samples):
        """imvoxel_occ_head.py:110-184 (semantic branch): {'loss_occ_i'} with level weight 0.5**i; seeds the logit gradients
        on the tape."""
        occ_preds = self.forward(x)
        gt_occ = [ds.gt_occupancy for ds in batch_data_samples]
        masks = None
        if getattr(batch_data_samples[0], 'gt_occupancy_masks', None) is not None:
            masks = [ds.gt_occupancy_masks for ds in batch_data_samples]
        C = self.num_classes
        losses = {}
        self.last = []
        for i, (logits, dims) in enumerate(occ_preds):
            dev = logits.d.device
            n = logits.d.shape[0]
            gt = self.targets(gt_occ, masks, 2 ** i, dims, dev)
            stats = torch.empty(3 * C + 2, dtype=torch.float64, device=dev)
            coeff = torch.empty(2 * C + 1, dtype=torch.float32, device=dev)
            out = torch.empty(4, dtype=torch.float32, device=dev)
            logits.g = torch.empty_like(logits.d)
            call('es_occ_loss', P(logits.d), logits.d.shape[1], P(gt), n, C, 0.5 ** i, P(stats), P(coeff), P(logits.g),
                 logits.g.shape[1], P(out), 0, hip.stream())
            losses[f'loss_occ_{i}'] = out[3]
            self.last.append(dict(logits=logits, gt=gt, parts=out, dims=dims))
        return losses
