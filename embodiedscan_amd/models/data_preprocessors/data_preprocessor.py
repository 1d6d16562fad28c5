"""Det3DDataPreprocessor for the MI355X path (embodiedscan/models/data_preprocessors/
data_preprocessor.py:132-339, multiview branch): one fused kernel does the BGR->RGB flip, float cast and
(x-mean)/std for all views of all samples and writes channels-last memory; the returned `imgs` tensor
keeps the reference's logical shape (B, V, 3, H, W) (a permuted view)."""
import torch
from ...hip import P, call, farr
from ...registry import MODELS


@MODELS.register_module()
class Det3DDataPreprocessor:
    def __init__(self, mean=None, std=None, bgr_to_rgb=False, rgb_to_bgr=False, pad_size_divisor=1, pad_value=0,
                 voxel=False, **kw):
        assert not voxel, 'voxel=True (mmcv hard/dynamic voxelisation) is not used by the shipped configs'
        self.mean, self.std = list(mean or [0, 0, 0]), list(std or [1, 1, 1])
        self.flip = bool(bgr_to_rgb or rgb_to_bgr)
        self.pad_size_divisor = pad_size_divisor
        self.device = torch.device('cuda:0')

    def forward(self, data, training=False):
        inputs, samples = data['inputs'], data.get('data_samples')
        out = {}
        if 'points' in inputs:
            out['points'] = [p.to(self.device, non_blocking=True) for p in inputs['points']]
        if 'img' in inputs or 'imgs' in inputs:
            img = inputs.get('img', inputs.get('imgs'))
            if isinstance(img, (list, tuple)):
                img = torch.stack([i.to(self.device, non_blocking=True) for i in img])
            img = img.to(self.device, non_blocking=True)
            B, V, C, H, W = img.shape
            assert img.dtype == torch.uint8 and C == 3
            assert H % self.pad_size_divisor == 0 and W % self.pad_size_divisor == 0, \
                'padding to the size divisor is not needed for the 480x480 inputs of the shipped configs'
            nhwc = torch.empty((B * V, H, W, 3), dtype=torch.float32, device=self.device)
            mean = self.mean if self.flip else self.mean[::-1]
            std = self.std if self.flip else self.std[::-1]
            # the kernel always reads channels [2,1,0]; without a flip feed it the mean/std reversed and un-flip
            assert self.flip, 'only the bgr_to_rgb=True configuration of the shipped configs is implemented'
            call('es_preprocess_img', P(img.contiguous()), B * V, H, W, farr(mean), farr(std), P(nhwc),
                 torch.cuda.current_stream().cuda_stream)
            out['imgs'] = nhwc.view(B, V, H, W, 3).permute(0, 1, 4, 2, 3)
            if samples is not None:
                for ds in samples:
                    ds.set_metainfo({'batch_input_shape': (H, W), 'pad_shape': (H, W)})
        return {'inputs': out, 'data_samples': samples}

    __call__ = forward
