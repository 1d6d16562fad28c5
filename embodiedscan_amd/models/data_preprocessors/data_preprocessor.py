"""Det3DDataPreprocessor for the MI355X path (embodiedscan/models/data_preprocessors/
data_preprocessor.py:132-339, multiview branch): one fused kernel does the BGR->RGB flip, float cast and
(x-mean)/std for all views of all samples and writes channels-last memory; the returned `imgs` tensor
keeps the reference's logical shape (B, V, 3, H, W) (a permuted view)."""
import torch
from ...hip import P, call, farr
from ...registry import MODELS


@MODELS.register_module()
class Det3DDataPreprocessor:
    RING = 2                                     # persistent output buffers per image geometry (see forward())

    def __init__(self, mean=None, std=None, bgr_to_rgb=False, rgb_to_bgr=False, pad_size_divisor=1, pad_value=0,
                 voxel=False, device=None, **kw):
        assert not voxel, 'voxel=True (mmcv hard/dynamic voxelisation) is not used by the shipped configs'
        self.mean, self.std = list(mean or [0, 0, 0]), list(std or [1, 1, 1])
        self.flip = bool(bgr_to_rgb or rgb_to_bgr)
        self.pad_size_divisor, self.pad_value = pad_size_divisor, pad_value
        # the owning detector sets this (constructor argument / detector.to()); None = follow the incoming tensors
        self.device = torch.device(device) if device is not None else None
        self._img_buf = {}

    def to(self, device):
        self.device = torch.device(device)
        return self

    def _target_device(self, inputs):
        if self.device is not None:
            return self.device
        for v in inputs.values():
            t = v[0] if isinstance(v, (list, tuple)) else v
            if torch.is_tensor(t) and t.is_cuda:
                return t.device
        return torch.device('cuda', torch.cuda.current_device())

    def forward(self, data, training=False):
        inputs, samples = data['inputs'], data.get('data_samples')
        out = {}
        dev = self._target_device(inputs)
        if 'points' in inputs:
            out['points'] = [p.to(dev, non_blocking=True) for p in inputs['points']]
        if 'img' in inputs or 'imgs' in inputs:
            img = inputs.get('img', inputs.get('imgs'))
            if isinstance(img, (list, tuple)):
                img = torch.stack([i.to(dev, non_blocking=True) for i in img])
            img = img.to(dev, non_blocking=True)
            B, V, C, H, W = img.shape
            assert img.dtype == torch.uint8 and C == 3
            d = max(int(self.pad_size_divisor), 1)
            Hp, Wp = (H + d - 1) // d * d, (W + d - 1) // d * d      # bottom / right padding (utils.py:43-62)
            # A RING of RING buffers per image geometry, rewritten in turn: the image backbone's launch sequence is replayed from a
            # hipGraph (one captured graph per input address) and so wants few, stable input addresses.  ALIASING CONTRACT: the
            # `imgs` returned by call n stay valid until call n + RING of the same geometry (round-3 advisor: with ONE buffer the
            # output of call n was overwritten by call n + 1, which breaks any caller that preprocesses the next batch while
            # the previous one is still live -- prefetching loops, tests comparing two batches); clone() to keep them longer.
            key = (B * V, Hp, Wp, str(dev))
            ring = self._img_buf.get(key)
            if ring is None:
                if len(self._img_buf) > 4:
                    self._img_buf.clear()
                ring = self._img_buf[key] = dict(bufs=[torch.empty((B * V, Hp, Wp, 3), dtype=torch.float32, device=dev)
                                                       for _ in range(self.RING)], n=0)
            nhwc = ring['bufs'][ring['n'] % self.RING]
            ring['n'] += 1
            call('es_preprocess_img', P(img.contiguous()), B * V, H, W, Hp, Wp, int(self.flip), farr(self.mean),
                 farr(self.std), float(self.pad_value), P(nhwc), torch.cuda.current_stream(dev).cuda_stream)
            out['imgs'] = nhwc.view(B, V, Hp, Wp, 3).permute(0, 1, 4, 2, 3)
            H, W = Hp, Wp
            if samples is not None:
                for ds in samples:
                    ds.set_metainfo({'batch_input_shape': (H, W), 'pad_shape': (H, W)})
        return {'inputs': out, 'data_samples': samples}

    __call__ = forward
