"""mmdet.ResNet (depth 50, torchvision-style bottlenecks, frozen eval-mode BN) on the es_hip row-matrix
convolution engine: images are channels-last row matrices (N*H*W, C); a 3x3 / strided conv is the sparse
conv kernel driven by a static image-grid kernel map, a 1x1 conv is a plain row GEMM.
Stand-in for the `backbone=dict(type='mmdet.ResNet', depth=50, base_channels=16, frozen_stages=1,
norm_eval=True, ...)` entry of configs/detection/mv-det3d_8xb4_embodiedscan-3d-284class-9dof.py:24-34,
called at embodiedscan/models/detectors/sparse_featfusion_single_stage.py:130-136.
"""
import os
import torch
from ... import engine as E
from ... import hip
from ...hip import P, call
from ...registry import MODELS

STEM_POOL = [os.environ.get('ES_STEM_POOL', '1') != '0']      # round 6: frozen stem + max pool as one launch (A/B switch)


def _stream():
    return hip.stream()


class _Grid:
    """static kernel maps of one (n_img, H, W) image grid."""

    def __init__(self, n_img, H, W, dev):
        self.n_img, self.H, self.W, self.dev = n_img, H, W, dev
        self.maps = {}

    def conv_map(self, kh, stride, pad, want_inv=True):
        key = (kh, stride, pad)
        if key not in self.maps:
            Ho, Wo = (self.H + 2 * pad - kh) // stride + 1, (self.W + 2 * pad - kh) // stride + 1
            n_out, K = self.n_img * Ho * Wo, kh * kh
            nbr = torch.empty((n_out, K), dtype=torch.int32, device=self.dev)
            call('es_image_map', self.n_img, self.H, self.W, Ho, Wo, kh, kh, stride, pad, P(nbr), _stream())
            inv = None
            if want_inv:
                n_in = self.n_img * self.H * self.W
                inv = torch.empty((n_in, K), dtype=torch.int32, device=self.dev)
                call('es_inverse_map', P(nbr), n_out, K, n_in, P(inv), _stream())
            self.maps[key] = (nbr, inv, n_out, Ho, Wo)
        m = self.maps[key]
        if hip.PROFILE is not None and m[0].data_ptr() not in hip.PAIRS:     # static maps: count pairs once
            hip.register_map(m[0])
            hip.register_map(m[1])
        return m


@MODELS.register_module(name='mmdet.ResNet')
class ResNet:
    arch = {50: (3, 4, 6, 3)}

    def __init__(self, depth=50, in_channels=3, stem_channels=None, base_channels=64, num_stages=4,
                 strides=(1, 2, 2, 2), out_indices=(0, 1, 2, 3), style='pytorch', frozen_stages=-1, norm_cfg=None,
                 norm_eval=True, init_cfg=None, **kw):
        assert depth == 50 and style == 'pytorch' and num_stages == 4 and tuple(strides) == (1, 2, 2, 2)
        assert norm_eval and norm_cfg is not None and not norm_cfg.get('requires_grad', True), \
            'only the frozen / eval-mode BN of the shipped config is implemented'
        self.base = base_channels
        self.out_indices, self.frozen_stages = tuple(out_indices), frozen_stages
        self.grids = {}
        # bf16 ACTIVATION storage (round 3; bf16 mode only): set by detectors whose only consumer of the feature maps is the
        # projection fusion (mv-3ddet, grounder); the occupancy detector's FPN still takes f32 rows
        self.act16 = False
        self._graphs, self._fold_version = {}, 0

    def bind(self, arena, prefix='backbone.'):
        self.arena, self.prefix = arena, prefix
        self._params = {}
        self.refresh()
        for k, v in arena.p.items():             # all conv kernels up front (one complete bf16 cast table from step 1)
            if k.startswith(prefix) and v.dim() == 3 and k != prefix + 'conv1.weight':
                self._par(k[len(prefix):])
        return self

    def _par(self, n):
        if n not in self._params:
            self._params[n] = E.Param(self.arena.p[self.prefix + n], self.arena.g.get(self.prefix + n))
        return self._params[n]

    def refresh(self):
        """fold every frozen BatchNorm2d into (scale, shift) -- call again after load_state_dict."""
        a, pre = self.arena, self.prefix
        self.fold = {}
        self._fold_version += 1                   # captured launch sequences hold pointers to the old constants
        for name in [k[len(pre):-len('.running_var')] for k in a.p if k.startswith(pre) and k.endswith('.running_var')]:
            C = a.p[pre + name + '.weight'].numel()
            sc = torch.empty(C, dtype=torch.float32, device=a.data.device)
            sh = torch.empty(C, dtype=torch.float32, device=a.data.device)
            call('es_bn_fold', P(a.p[pre + name + '.weight']), P(a.p[pre + name + '.bias']),
                 P(a.p[pre + name + '.running_mean']), P(a.p[pre + name + '.running_var']), C, 1e-5, P(sc), P(sh),
                 _stream())
            self.fold[name] = (sc, sh)

    def forward(self, x):
        """x: (n_img, H, W, 3) f32 channels-last.  Returns [(Var (n_img*h*w, C), h, w)] for the out_indices."""
        n_img, H, W, _ = x.shape
        dev = x.device
        a16 = bool(self.act16 and E.ACT16[0] and E.PRECISION[0] == 'bf16' and self.frozen_stages >= 0)
        fused = self.base in (16, 32, 64) and self.frozen_stages >= 0 and a16 and STEM_POOL[0]     # stem + max pool in one launch
        direct = (self.base in (16, 32) and self.frozen_stages >= 0) or fused
        key = (n_img, H, W, direct)
        if key not in self.grids:
            if direct:                          # fused stem kernel: no 49-tap image map needed
                Ho, Wo = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
                stem, nbr_a, nbr_b = (None, None, n_img * Ho * Wo, Ho, Wo), None, None
            else:
                stem = _Grid(n_img, H, W, dev).conv_map(7, 2, 3, want_inv=False)
                nbr_a, nbr_b = stem[0][:, :27].contiguous(), stem[0][:, 27:].contiguous()
            g1 = _Grid(n_img, stem[3], stem[4], dev)
            pool = g1.conv_map(3, 2, 1, want_inv=False)
            grids = [_Grid(n_img, pool[3], pool[4], dev)]
            for li in range(1, 4):
                h, w = grids[-1].H, grids[-1].W
                grids.append(_Grid(n_img, (h + 2 - 3) // 2 + 1, (w + 2 - 3) // 2 + 1, dev))
            self.grids[key] = (stem, nbr_a, nbr_b, pool, grids)
        stem, nbr_a, nbr_b, pool, grids = self.grids[key]
        s = _stream()
        prev = E.TAPE.enabled
        E.TAPE.enabled = prev and self.frozen_stages < 0
        # ---- stem: direct 7x7 s2 conv + frozen BN + ReLU (one fused kernel), then 3x3 s2 max pool
        w1 = self.arena.p[self.prefix + 'conv1.weight']
        xin = x.reshape(n_img * H * W, 3)
        if fused:
            # round 6: stem + max pool in one launch, bf16 rows out (bit-identical to the pair below)
            yp = torch.empty((pool[2], self.base), dtype=torch.bfloat16, device=dev)
            call('es_stem_pool_fwd', P(xin), P(w1), P(self.fold['bn1'][0]), P(self.fold['bn1'][1]), n_img, H, W, self.base, P(yp), s)
            cur = None
        elif self.base in (16, 32) and self.frozen_stages >= 0:
            y = torch.empty((stem[2], self.base), dtype=torch.float32, device=dev)
            call('es_stem_conv_fwd', P(xin), P(w1), P(self.fold['bn1'][0]), P(self.fold['bn1'][1]), n_img, H, W,
                 self.base, P(y), s)
            cur = E.Var(y, rg=False)
        else:                                   # generic engine: 49 taps as 27 + 22
            y = torch.empty((stem[2], self.base), dtype=torch.float32, device=dev)
            call('es_spconv_fwd', P(xin), 3, P(w1), P(nbr_a), stem[2], xin.shape[0], 27, 3, self.base, 0, P(y),
                 self.base, 0, 0, s)
            call('es_spconv_fwd', P(xin), 3, w1.data_ptr() + 4 * 27 * 3 * self.base, P(nbr_b), stem[2], xin.shape[0], 22,
                 3, self.base, 0, P(y), self.base, 0, 1, s)
            cur = E.affine_act(E.Var(y, rg=False), *self.fold['bn1'], act=1)
        if cur is None:
            cur = E.Var(yp, rg=False)
            cur.dh = yp
        elif a16:                               # frozen stem: forward-only pooling straight into bf16 rows
            yp = torch.empty((pool[2], self.base), dtype=torch.bfloat16, device=dev)
            call('es_maxpool_fwd_h', P(cur.d), self.base, P(pool[0]), pool[2], pool[0].shape[1], self.base, P(yp), s)
            cur = E.Var(yp, rg=False)
            cur.dh = yp
        else:
            cur = E.maxpool(cur, pool[0], pool[2], need_dx=False)
        cur.rg = self.frozen_stages < 0
        outs = []
        for li, nblk in enumerate(self.arch[50]):
            E.TAPE.enabled = prev and (li + 1) > self.frozen_stages
            gin = grids[li - 1] if li > 0 else grids[0]
            for bi in range(nblk):
                p = f'layer{li + 1}.{bi}.'
                stride = 2 if (bi == 0 and li > 0) else 1
                g_in = gin if bi == 0 else grids[li]
                o = E.conv_affine(cur, self._par(p + 'conv1.weight'), None, None, cur.d.shape[0], *self.fold[p + 'bn1'],
                                  act=1, out_bf16=a16)
                nbr, inv, n_out, _, _ = g_in.conv_map(3, stride, 1)
                o = E.conv_affine(o, self._par(p + 'conv2.weight'), nbr, inv, n_out, *self.fold[p + 'bn2'], act=1,
                                  sole_consumer=True, out_bf16=a16, img=(n_img, g_in.H, g_in.W, stride))
                if bi == 0:
                    if stride == 1:
                        idt = E.conv_affine(cur, self._par(p + 'downsample.0.weight'), None, None, n_out,
                                            *self.fold[p + 'downsample.1'], act=0, out_bf16=a16)
                    else:
                        dn, di, _, _, _ = g_in.conv_map(1, stride, 0)
                        idt = E.conv_affine(cur, self._par(p + 'downsample.0.weight'), dn, di, n_out,
                                            *self.fold[p + 'downsample.1'], act=0, out_bf16=a16)
                else:
                    idt = cur
                cur = E.conv_affine(o, self._par(p + 'conv3.weight'), None, None, n_out, *self.fold[p + 'bn3'], act=1,
                                    res=idt, sole_consumer=True, out_bf16=a16)
                if not E.TAPE.enabled:
                    cur.rg = False
            if li in self.out_indices:
                outs.append((cur, grids[li].H, grids[li].W))
        E.TAPE.enabled = prev
        return outs

    def __call__(self, x):
        """forward(x) through engine.graphed(): the launch sequence is static for a given input buffer, grid and mode"""
        # the table-driven bf16 weight cast must run OUTSIDE the captured sequence (inside, capture would stamp every kernel as
        # cast without executing the launch, and every replay would redo it): bring the copies up to date first
        E.refresh_weight_copies()
        key = (x.data_ptr(), tuple(x.shape), E.PRECISION[0], bool(self.act16 and E.ACT16[0]), E.TAPE.enabled, self._fold_version,
               self.arena.data.data_ptr(), E.SHADOW[0], E.WGRAD_SHADOW[0], E.IMG_CONV[0])
        return E.graphed(self._graphs, key, lambda: self.forward(x))
