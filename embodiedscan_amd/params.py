"""Parameter inventory of the mv-3ddet detector under the reference's state-dict names.

One flat f32 buffer holds every parameter (and a second one every gradient), so that
gradient all-reduce, grad-norm clipping and AdamW are each ONE pass over HBM
(DESIGN.md "flat parameter arena").  Names / shapes follow:
  * mmdet.ResNet(depth=50, base_channels=16)   configs/detection/mv-det3d_...py:24-34
  * MinkResNet(depth=34, in_channels=3)        embodiedscan/models/backbones/mink_resnet.py:58-120
  * FCAF3DHeadRotMat                           embodiedscan/models/dense_heads/fcaf3d_head.py:949-991
Sparse conv kernels use the MinkowskiEngine layout [K^3, C_in, C_out] ([C_in, C_out] for k=1).
"""
import math
from collections import OrderedDict
import torch


class Spec:
    """One arena tensor.  `ref` describes how it appears in the reference state dict:
    None = same name / shape; ('oihw', (O,I,KH,KW)) = a conv2d weight kept as [KH*KW, I, O] for the
    row-matrix conv engine; ('head_out',) = the fused 1x1 head kernels (see fcaf3d_head_specs)."""
    __slots__ = ('name', 'shape', 'init', 'trainable', 'buffer', 'ref', 'aliases')

    def __init__(self, name, shape, init, trainable=True, buffer=False, ref=None, aliases=()):
        self.name, self.shape, self.init, self.trainable, self.buffer = name, tuple(shape), init, trainable, buffer
        self.ref = ref
        # further reference keys holding the SAME tensor (modules shared through an nn.ModuleList, e.g. GroundingHead's
        # share_pred_layer=True branches): emitted as copies, absorbed from whichever key is present
        self.aliases = tuple(aliases)


def _conv2d(name, o, i, kh, kw, fan, trainable):
    return Spec(name, (kh * kw, i, o), ('kaiming_out', fan), trainable, ref=('oihw', (o, i, kh, kw)))


def _bn2d(specs, p, c):
    specs += [Spec(p + '.weight', (c,), ('const', 1.), False), Spec(p + '.bias', (c,), ('const', 0.), False),
              Spec(p + '.running_mean', (c,), ('const', 0.), False, True),
              Spec(p + '.running_var', (c,), ('const', 1.), False, True)]


def resnet50_specs(prefix='backbone.', base=16, frozen_stages=1):
    s = []
    s.append(_conv2d(prefix + 'conv1.weight', base, 3, 7, 7, base * 49, frozen_stages < 0))
    _bn2d(s, prefix + 'bn1', base)
    inpl = base
    for li, nblk in enumerate((3, 4, 6, 3)):
        planes = base * 2 ** li
        train = (li + 1) > frozen_stages
        for bi in range(nblk):
            p = f'{prefix}layer{li + 1}.{bi}.'
            s.append(_conv2d(p + 'conv1.weight', planes, inpl, 1, 1, planes, train))
            _bn2d(s, p + 'bn1', planes)
            s.append(_conv2d(p + 'conv2.weight', planes, planes, 3, 3, planes * 9, train))
            _bn2d(s, p + 'bn2', planes)
            s.append(_conv2d(p + 'conv3.weight', planes * 4, planes, 1, 1, planes * 4, train))
            _bn2d(s, p + 'bn3', planes * 4)
            if bi == 0:
                s.append(_conv2d(p + 'downsample.0.weight', planes * 4, inpl, 1, 1, planes * 4, train))
                _bn2d(s, p + 'downsample.1', planes * 4)
            inpl = planes * 4
    return s


def _mbn(specs, p, c):
    specs += [Spec(p + '.bn.weight', (c,), ('const', 1.)), Spec(p + '.bn.bias', (c,), ('const', 0.)),
              Spec(p + '.bn.running_mean', (c,), ('const', 0.), False, True),
              Spec(p + '.bn.running_var', (c,), ('const', 1.), False, True)]


def mink_resnet34_specs(prefix='backbone_3d.', in_channels=3):
    s = [Spec(prefix + 'conv1.kernel', (27, in_channels, 64), ('kaiming_out', 27 * 64)),
         Spec(prefix + 'norm1.weight', (1, 64), ('const', 1.)), Spec(prefix + 'norm1.bias', (1, 64), ('const', 0.))]
    inpl = 64
    for li, nblk in enumerate((3, 4, 6, 3)):
        planes = 64 * 2 ** li
        for bi in range(nblk):
            p = f'{prefix}layer{li + 1}.{bi}.'
            s.append(Spec(p + 'conv1.kernel', (27, inpl, planes), ('kaiming_out', 27 * planes)))
            _mbn(s, p + 'norm1', planes)
            s.append(Spec(p + 'conv2.kernel', (27, planes, planes), ('kaiming_out', 27 * planes)))
            _mbn(s, p + 'norm2', planes)
            if bi == 0:
                s.append(Spec(p + 'downsample.0.kernel', (1, inpl, planes), ('kaiming_out', planes), ref=('squeeze0',)))
                _mbn(s, p + 'downsample.1', planes)
            inpl = planes
    return s


def fcaf3d_head_specs(prefix='bbox_head.', in_channels=(128, 256, 512, 1024), out_channels=128, n_reg=12,
                      n_classes=284):
    s = []
    for i, c in enumerate(in_channels):
        if i > 0:
            p = f'{prefix}up_block_{i}'
            co = in_channels[i - 1]
            s.append(Spec(p + '.0.kernel', (8, c, co), ('uniform_fan', co * 8)))     # transpose: n = out*vol
            _mbn(s, p + '.1', co)
            s.append(Spec(p + '.3.kernel', (27, co, co), ('uniform_fan', co * 27)))
            _mbn(s, p + '.4', co)
        p = f'{prefix}out_block_{i}'
        s.append(Spec(p + '.0.kernel', (27, c, out_channels), ('uniform_fan', c * 27)))
        _mbn(s, p + '.1', out_channels)
    # conv_center (C,1) | conv_reg (C,n_reg) | conv_cls (C,n_classes) fused into ONE row GEMM; exported to the
    # reference names conv_center.kernel / conv_reg.kernel / conv_cls.kernel / conv_cls.bias
    # columns padded to a multiple of 64 so that the head GEMM, its data gradient and its weight gradient run on the fast
    # (unchecked) bf16 kernels: 297 -> 320; the padding columns are zero, receive zero gradients and are never read
    nh = 1 + n_reg + n_classes
    nhp = (nh + 63) // 64 * 64
    s.append(Spec(prefix + 'head_out.kernel', (1, out_channels, nhp), ('normal_pad', (.01, nh)), ref=('head_out', n_reg, n_classes)))
    s.append(Spec(prefix + 'head_out.bias', (nhp,), ('head_bias', (1 + n_reg, -math.log((1 - .01) / .01), nh)),
                  ref=('head_bias', n_reg, n_classes)))
    for i in range(len(in_channels)):
        s.append(Spec(f'{prefix}scales.{i}.scale', (), ('const', 1.)))
    return s


def detector_specs(n_classes=284):
    return resnet50_specs() + mink_resnet34_specs() + fcaf3d_head_specs(n_classes=n_classes)


# ---------------------------------------------------------------- grounding path (BASELINE config 4)
def _linear(name, cin, cout, init=None, trainable=True, aliases=()):
    return Spec(name, (1, cin, cout), init or ('uniform_fan', cin), trainable, ref=('linear',), aliases=aliases)


def mink_neck_specs(prefix='neck_3d.', in_channels=(128, 256, 512, 1024), out_channels=256, num_classes=1):
    """MinkNeck (embodiedscan/models/necks/mink_neck.py:46-131): the FCAF3D-style sparse FPN + a 1x1 score conv"""
    s = []
    for i, c in enumerate(in_channels):
        if i > 0:
            p = f'{prefix}up_block_{i}'
            co = in_channels[i - 1]
            s.append(Spec(p + '.0.kernel', (8, c, co), ('uniform_fan', co * 8)))
            _mbn(s, p + '.1', co)
            s.append(Spec(p + '.3.kernel', (27, co, co), ('uniform_fan', co * 27)))
            _mbn(s, p + '.4', co)
        p = f'{prefix}out_block_{i}'
        s.append(Spec(p + '.0.kernel', (27, c, out_channels), ('uniform_fan', c * 27)))
        _mbn(s, p + '.1', out_channels)
    # conv_cls only ranks voxels for pruning (under no_grad in the reference): it never receives a gradient
    s.append(Spec(prefix + 'conv_cls.kernel', (1, out_channels, num_classes), ('normal', .01), False, ref=('squeeze0',)))
    s.append(Spec(prefix + 'conv_cls.bias', (1, num_classes), ('const', -math.log((1 - .01) / .01)), False))
    return s


def _posembed_specs(s, p, cin, E, trainable=True):
    """PositionEmbeddingLearned (ground_transformer/decoder.py:20-34): Conv1d(cin,E,1) BN1d ReLU Conv1d(E,E,1)"""
    q = p + '.position_embedding_head'
    s.append(Spec(q + '.0.weight', (1, cin, E), ('uniform_fan', cin), trainable, ref=('conv1d',)))
    s.append(Spec(q + '.0.bias', (E,), ('uniform_fan', cin), trainable))
    s += [Spec(q + '.1.weight', (E,), ('const', 1.), trainable), Spec(q + '.1.bias', (E,), ('const', 0.), trainable),
          Spec(q + '.1.running_mean', (E,), ('const', 0.), False, True), Spec(q + '.1.running_var', (E,), ('const', 1.), False, True)]
    s.append(Spec(q + '.3.weight', (1, E, E), ('uniform_fan', E), trainable, ref=('conv1d',)))
    s.append(Spec(q + '.3.bias', (E,), ('uniform_fan', E), trainable))


def ground_decoder_specs(prefix='decoder.', num_layers=6, E=256, ffn=2048):
    """SparseFeatureFusionTransformerDecoder (decoder.py:182-297) with mmcv MultiheadAttention / FFN key names"""
    s = []
    for i in range(num_layers):
        p = f'{prefix}layers.{i}.'
        for a in ('self_attn', 'cross_attn_text', 'cross_attn'):
            s.append(Spec(p + a + '.attn.in_proj_weight', (3, E, E), ('xavier', (E, 3 * E)), ref=('inproj',)))
            s.append(Spec(p + a + '.attn.in_proj_bias', (3, E), ('const', 0.), ref=('reshape', (3 * E,))))
            s.append(_linear(p + a + '.attn.out_proj.weight', E, E))
            s.append(Spec(p + a + '.attn.out_proj.bias', (E,), ('const', 0.)))
        s.append(_linear(p + 'ffn.layers.0.0.weight', E, ffn))
        s.append(Spec(p + 'ffn.layers.0.0.bias', (ffn,), ('uniform_fan', E)))
        s.append(_linear(p + 'ffn.layers.1.weight', ffn, E))
        s.append(Spec(p + 'ffn.layers.1.bias', (E,), ('uniform_fan', ffn)))
        for k in range(4):
            s += [Spec(p + f'norms.{k}.weight', (E,), ('const', 1.)), Spec(p + f'norms.{k}.bias', (E,), ('const', 0.))]
        # the per-layer self_posembed exists in the reference module but its forward never uses it (decoder.py:98,
        # 267-271 use the decoder-level embeddings): parameters without a gradient, kept frozen
        _posembed_specs(s, p + 'self_posembed', 3, E, trainable=False)
    _posembed_specs(s, prefix + 'self_posembed', 9, E)
    _posembed_specs(s, prefix + 'cross_posembed', 3, E)
    s += [Spec(prefix + 'norm.weight', (E,), ('const', 1.)), Spec(prefix + 'norm.bias', (E,), ('const', 0.))]
    return s


def grounding_head_specs(prefix='bbox_head.', E=256, num_reg=9, num_pred_layer=7):
    """GroundingHead with share_pred_layer=True (grounding_head.py:188-224): ONE ContrastiveEmbed bias and ONE
    Linear-ReLU-Linear-ReLU-Linear regression branch, visible under 7 ModuleList indices in the reference state dict"""
    al = lambda fmt: [prefix + fmt.format(i) for i in range(1, num_pred_layer)]
    s = [Spec(prefix + 'cls_branches.0.bias', (1,), ('const', -math.log((1 - 0.01) / 0.01)), aliases=al('cls_branches.{}.bias'))]
    for j, (ci, co) in zip((0, 2, 4), ((E, E), (E, E), (E, num_reg))):
        last = j == 4
        s.append(_linear(prefix + f'reg_branches.0.{j}.weight', ci, co, ('const', 0.) if last else None,
                         aliases=al('reg_branches.{}.' + f'{j}.weight')))
        s.append(Spec(prefix + f'reg_branches.0.{j}.bias', (co,), ('reg_bias', -2.0) if last else ('uniform_fan', ci),
                      aliases=al('reg_branches.{}.' + f'{j}.bias')))
    return s


def grounder_specs(text_dim=768, E=256, num_layers=6, ffn=2048, in_channels=(128, 256, 512, 1024)):
    return (resnet50_specs() + mink_resnet34_specs() + mink_neck_specs(in_channels=in_channels, out_channels=E) +
            ground_decoder_specs(num_layers=num_layers, E=E, ffn=ffn) +
            [_linear('text_feat_map.weight', text_dim, E), Spec('text_feat_map.bias', (E,), ('uniform_fan', text_dim))] +
            grounding_head_specs(E=E, num_pred_layer=num_layers + 1))


# ---------------------------------------------------------------- occupancy path (BASELINE config 5)
def _conv3d(name, o, i, k, stride_fan=None):
    """nn.Conv3d weight (O, I, k, k, k), default PyTorch init (kaiming_uniform a=sqrt(5) == U(+-1/sqrt(fan_in)))"""
    return Spec(name, (k ** 3, i, o), ('uniform_fan', i * k ** 3), ref=('oidhw', (o, i, k, k, k)))


def _bn3d(specs, p, c):
    specs += [Spec(p + '.weight', (c,), ('const', 1.)), Spec(p + '.bias', (c,), ('const', 0.)),
              Spec(p + '.running_mean', (c,), ('const', 0.), False, True),
              Spec(p + '.running_var', (c,), ('const', 1.), False, True)]


def fpn_specs(prefix='neck.', in_channels=(256, 512, 1024, 2048), out_channels=256):
    """mmdet.FPN: lateral_convs.i.conv (1x1, bias), fpn_convs.i.conv (3x3, bias); xavier-uniform init"""
    s = []
    for i, c in enumerate(in_channels):
        s.append(Spec(f'{prefix}lateral_convs.{i}.conv.weight', (1, c, out_channels), ('xavier', (c, out_channels)),
                      ref=('oihw', (out_channels, c, 1, 1))))
        s.append(Spec(f'{prefix}lateral_convs.{i}.conv.bias', (out_channels,), ('const', 0.)))
    for i in range(len(in_channels)):
        s.append(Spec(f'{prefix}fpn_convs.{i}.conv.weight', (9, out_channels, out_channels),
                      ('xavier', (out_channels * 9, out_channels * 9)), ref=('oihw', (out_channels, out_channels, 3, 3))))
        s.append(Spec(f'{prefix}fpn_convs.{i}.conv.bias', (out_channels,), ('const', 0.)))
    return s


def imvoxel_neck_specs(prefix='neck_3d.', in_channels=768, out_channels=128, n_blocks=(1, 1, 1)):
    """IndoorImVoxelNeck (embodiedscan/models/necks/imvoxel_neck.py:19-143) under its state-dict names"""
    s = []
    c = in_channels

    def res(p, ci, co, stride):
        s.append(_conv3d(p + '.conv1.weight', co, ci, 3))
        _bn3d(s, p + '.norm1', co)
        s.append(_conv3d(p + '.conv2.weight', co, co, 3))
        _bn3d(s, p + '.norm2', co)
        if stride != 1:
            s.append(_conv3d(p + '.downsample.0.weight', co, ci, 1))
            _bn3d(s, p + '.downsample.1', co)

    for i, nb in enumerate(n_blocks):
        stride = 1 if i == 0 else 2
        for b in range(nb):
            if b == 0 and stride != 1:
                res(f'{prefix}down_layer_{i}.{b}', c, c * 2, stride)
                c *= 2
            else:
                res(f'{prefix}down_layer_{i}.{b}', c, c, 1)
        if i > 0:
            p = f'{prefix}up_block_{i}'
            # nn.ConvTranspose3d weight (I, O, 2, 2, 2); PyTorch's fan_in for it is size(1) * 8
            s.append(Spec(p + '.0.weight', (8, c, c // 2), ('uniform_fan', (c // 2) * 8), ref=('iodhw', (c, c // 2, 2, 2, 2))))
            _bn3d(s, p + '.1', c // 2)
            s.append(_conv3d(p + '.3.weight', c // 2, c // 2, 3))
            _bn3d(s, p + '.4', c // 2)
        p = f'{prefix}out_block_{i}'
        s.append(_conv3d(p + '.0.weight', out_channels, c, 3))
        _bn3d(s, p + '.1', out_channels)
    return s


def occ_head_specs(prefix='bbox_head.', in_channels=(128, 128, 128), num_classes=81):
    return [_conv3d(f'{prefix}occ.{i}.weight', num_classes, c, 1) for i, c in enumerate(in_channels)]


def occ_detector_specs(base_channels=64, fpn_out=256, neck_in=768, neck_out=128, n_blocks=(1, 1, 1), num_classes=81,
                       head_in=(128, 128, 128)):
    b = base_channels
    # backbones first, then everything else: the arena order defines the gradient buckets (parallel.BucketedGradReducer)
    return (resnet50_specs(base=b) + mink_resnet34_specs() +
            fpn_specs(in_channels=(4 * b, 8 * b, 16 * b, 32 * b), out_channels=fpn_out) + imvoxel_neck_specs(in_channels=neck_in, out_channels=neck_out, n_blocks=n_blocks) +
            occ_head_specs(in_channels=head_in, num_classes=num_classes))


def _fill(t, init, gen):
    kind, a = init
    if kind == 'const':
        t.fill_(a)
    elif kind == 'kaiming_out':
        t.normal_(0, math.sqrt(2.0 / a), generator=gen)
    elif kind == 'uniform_fan':
        b = 1.0 / math.sqrt(a)
        t.uniform_(-b, b, generator=gen)
    elif kind == 'xavier':
        b = math.sqrt(6.0 / (a[0] + a[1]))
        t.uniform_(-b, b, generator=gen)
    elif kind == 'normal':
        t.normal_(0, a, generator=gen)
    elif kind == 'reg_bias':                # GroundingHead.init_weights: last reg layer bias 0, bias[2:] = -2 (grounding_head.py:220-224)
        t.zero_()
        t[2:] = a
    elif kind == 'normal_pad':              # N(0, std) in the first a[1] columns of the last dim, zero padding behind
        # (draws exactly the numbers an un-padded tensor would: the random stream of the other parameters is unchanged)
        live = torch.empty(t.shape[:-1] + (a[1],), dtype=t.dtype).normal_(0, a[0], generator=gen)
        t.zero_()
        t[..., :a[1]] = live
    elif kind == 'head_bias':
        t.zero_()
        t[a[0]:(a[2] if len(a) > 2 else None)] = a[1]
    else:
        raise ValueError(kind)


class ParamArena:
    """Flat parameter / gradient storage with named views.

    Trainable tensors are packed first (``data[:n_train]`` / ``grad``), frozen
    parameters and buffers after them, every tensor 16-byte aligned."""

    def __init__(self, specs, seed=0, device='cpu'):
        self.specs = specs
        order = [s for s in specs if s.trainable] + [s for s in specs if not s.trainable]
        off, self.offsets = 0, {}
        for s in order:
            n = 1
            for d in s.shape:
                n *= d
            self.offsets[s.name] = (off, n)
            off += (n + 3) // 4 * 4
            if s.trainable:
                self.n_train = off
        self.total = off
        gen = torch.Generator().manual_seed(seed)
        data = torch.zeros(self.total, dtype=torch.float32)
        for s in specs:                     # init in spec order so the stream is layout independent
            o, n = self.offsets[s.name]
            _fill(data[o:o + n].view(s.shape), s.init, gen)
        self.data = data.to(device)
        self.grad = torch.zeros(self.n_train, dtype=torch.float32, device=device)
        self._views()

    def _views(self):
        self.p, self.g = OrderedDict(), OrderedDict()
        for s in self.specs:
            o, n = self.offsets[s.name]
            self.p[s.name] = self.data[o:o + n].view(s.shape)
            if s.trainable:
                self.g[s.name] = self.grad[o:o + n].view(s.shape)

    def to(self, device):
        self.data = self.data.to(device)
        self.grad = self.grad.to(device)
        self._views()
        return self

    # ---- reference <-> arena layout conversions, shared by parameters, gradients and optimiser moments
    @staticmethod
    def _emit(s, t, out):
        """arena-layout tensor t of spec s -> its reference-named / reference-shaped entries in `out`"""
        if s.ref is None:
            out[s.name] = t.clone()
        elif s.ref[0] == 'oihw':
            o, i, kh, kw = s.ref[1]
            out[s.name] = t.reshape(kh, kw, i, o).permute(3, 2, 0, 1).contiguous()
        elif s.ref[0] == 'squeeze0':
            out[s.name] = t[0].clone()
        elif s.ref[0] == 'linear':                     # arena [1][in][out] -> nn.Linear (out, in)
            out[s.name] = t[0].t().contiguous()
        elif s.ref[0] == 'conv1d':                     # arena [1][in][out] -> nn.Conv1d (out, in, 1)
            out[s.name] = t[0].t().contiguous().unsqueeze(-1)
        elif s.ref[0] == 'inproj':                     # arena [3][in][out] (q, k, v) -> in_proj_weight (3*out, in)
            out[s.name] = t.transpose(1, 2).reshape(-1, t.shape[1]).contiguous()
        elif s.ref[0] == 'reshape':
            out[s.name] = t.reshape(s.ref[1]).clone()
        elif s.ref[0] == 'oidhw':                      # arena [kd*kh*kw][I][O] -> nn.Conv3d (O, I, kd, kh, kw)
            o, i, kd, kh, kw = s.ref[1]
            out[s.name] = t.reshape(kd, kh, kw, i, o).permute(4, 3, 0, 1, 2).contiguous()
        elif s.ref[0] == 'iodhw':                      # arena [kd*kh*kw][I][O] -> nn.ConvTranspose3d (I, O, kd, kh, kw)
            i, o, kd, kh, kw = s.ref[1]
            out[s.name] = t.reshape(kd, kh, kw, i, o).permute(3, 4, 0, 1, 2).contiguous()
        elif s.ref[0] == 'head_out':
            pre = s.name[:-len('head_out.kernel')]
            nr = s.ref[1]
            out[pre + 'conv_center.kernel'] = t[0, :, 0:1].clone()
            out[pre + 'conv_reg.kernel'] = t[0, :, 1:1 + nr].clone()
            out[pre + 'conv_cls.kernel'] = t[0, :, 1 + nr:1 + nr + s.ref[2]].clone()
        elif s.ref[0] == 'head_bias':
            pre = s.name[:-len('head_out.bias')]
            out[pre + 'conv_cls.bias'] = t[1 + s.ref[1]:1 + s.ref[1] + s.ref[2]].reshape(1, -1).clone()

    @staticmethod
    def _absorb(s, dst, sd):
        """inverse of _emit: copy the entries of spec s found in `sd` into the arena-layout view dst; returns the
        reference names it consumed"""
        if s.ref is None or s.ref[0] == 'squeeze0':
            if s.name in sd:
                dst.copy_(sd[s.name].reshape(dst.shape))
                return [s.name]
        elif s.ref[0] == 'oihw':
            if s.name in sd:
                o, i, kh, kw = s.ref[1]
                dst.copy_(sd[s.name].permute(2, 3, 1, 0).reshape(kh * kw, i, o))
                return [s.name]
        elif s.ref[0] in ('linear', 'conv1d'):
            if s.name in sd:
                dst[0].copy_(sd[s.name].reshape(dst.shape[2], dst.shape[1]).t())
                return [s.name]
        elif s.ref[0] == 'inproj':
            if s.name in sd:
                dst.copy_(sd[s.name].reshape(3, dst.shape[2], dst.shape[1]).transpose(1, 2))
                return [s.name]
        elif s.ref[0] == 'reshape':
            if s.name in sd:
                dst.copy_(sd[s.name].reshape(dst.shape))
                return [s.name]
        elif s.ref[0] == 'oidhw':
            if s.name in sd:
                o, i, kd, kh, kw = s.ref[1]
                dst.copy_(sd[s.name].permute(2, 3, 4, 1, 0).reshape(kd * kh * kw, i, o))
                return [s.name]
        elif s.ref[0] == 'iodhw':
            if s.name in sd:
                i, o, kd, kh, kw = s.ref[1]
                dst.copy_(sd[s.name].permute(2, 3, 4, 0, 1).reshape(kd * kh * kw, i, o))
                return [s.name]
        elif s.ref[0] == 'head_out':
            pre = s.name[:-len('head_out.kernel')]
            nr = s.ref[1]
            names = [pre + 'conv_center.kernel', pre + 'conv_reg.kernel', pre + 'conv_cls.kernel']
            if all(n in sd for n in names):
                dst[0, :, 0:1].copy_(sd[names[0]])
                dst[0, :, 1:1 + nr].copy_(sd[names[1]])
                dst[0, :, 1 + nr:1 + nr + s.ref[2]].copy_(sd[names[2]])
                return names
        elif s.ref[0] == 'head_bias':
            pre = s.name[:-len('head_out.bias')]
            if pre + 'conv_cls.bias' in sd:
                dst[1 + s.ref[1]:1 + s.ref[1] + s.ref[2]].copy_(sd[pre + 'conv_cls.bias'].reshape(-1))
                return [pre + 'conv_cls.bias']
        return []

    @classmethod
    def _emit_all(cls, s, t, out):
        cls._emit(s, t, out)
        for a in s.aliases:
            out[a] = out[s.name].clone()

    @classmethod
    def _absorb_all(cls, s, dst, sd):
        got = cls._absorb(s, dst, sd)
        if not got:
            for a in s.aliases:
                if a in sd:
                    cls._absorb(s, dst, {s.name: sd[a]})
                    got = [a]
                    break
        return got + [a for a in s.aliases if a in sd and a not in got] if got else got

    def ref_names(self, s):
        """reference key(s) spec s maps to"""
        probe = OrderedDict()
        self._emit(s, self.p[s.name].detach(), probe)
        return list(probe)

    def state_dict(self):
        """Reference-named, reference-shaped copy (what `model.state_dict()` gives in the reference)."""
        out = OrderedDict()
        for s in self.specs:
            self._emit_all(s, self.p[s.name].detach(), out)
        return out

    def _to_ref(self, which):
        """reference-named view of the gradient (which='g') arena, same conversions as state_dict()."""
        src = self.g if which == 'g' else self.p
        out = OrderedDict()
        for s in self.specs:
            if s.name in src:
                self._emit(s, src[s.name].detach(), out)
        return out

    def grad_dict(self):
        return self._to_ref('g')

    def flat_to_ref(self, flat):
        """a flat buffer laid out like the TRAINABLE part of the arena (optimiser moments) -> reference-named dict"""
        out = OrderedDict()
        for s in self.specs:
            if s.trainable:
                o, n = self.offsets[s.name]
                self._emit(s, flat[o:o + n].view(s.shape).detach(), out)
        return out

    def ref_to_flat(self, sd, flat):
        """inverse of flat_to_ref, in place"""
        for s in self.specs:
            if s.trainable:
                o, n = self.offsets[s.name]
                self._absorb(s, flat[o:o + n].view(s.shape), sd)
        return flat

    @staticmethod
    def tap_permutation(order, ksize):
        """index list p with arena_kernel[k] = checkpoint_kernel[p[k]] for a sparse ksize^3 kernel.
        The arena (and the oracle) number the taps with x fastest: k = (iz*ks + iy)*ks + ix (ParamArena docstring).
        MinkowskiEngine's own region iterator order cannot be verified offline, so a checkpoint written by the
        reference may use another numbering: 'x_fastest' (identity), 'z_fastest' (k' = (ix*ks + iy)*ks + iz), or an
        explicit list of ks^3 indices."""
        n = ksize ** 3
        if order in (None, 'x_fastest'):
            return list(range(n))
        if order == 'z_fastest':
            return [((k % ksize) * ksize + (k // ksize) % ksize) * ksize + k // (ksize * ksize) for k in range(n)]
        order = list(order)
        assert sorted(order) == list(range(n)), f'tap order must be a permutation of range({n})'
        return order

    def load_state_dict(self, sd, strict=False, tap_order=None):
        """Accepts a reference-named state dict (the inverse of state_dict()).  Returns (missing, unexpected)
        reference keys; strict=True raises if either is non-empty (torch.nn.Module.load_state_dict semantics).
        tap_order: numbering of the 3^3 / 2^3 kernel offsets in the CHECKPOINT's sparse kernels ('x_fastest' = ours,
        'z_fastest', or {27: [...], 8: [...]} explicit permutations); the kernels are re-ordered while loading."""
        used, missing = set(), []
        perms = {}
        if tap_order not in (None, 'x_fastest'):
            for ks in (3, 2):
                perms[ks ** 3] = self.tap_permutation(tap_order.get(ks ** 3) if isinstance(tap_order, dict) else tap_order, ks)
        for s in self.specs:
            got = self._absorb_all(s, self.p[s.name], sd)
            if got and perms and s.ref is None and s.name.endswith('.kernel') and len(s.shape) == 3 and s.shape[0] in perms:
                self.p[s.name].copy_(self.p[s.name][torch.tensor(perms[s.shape[0]], device=self.p[s.name].device)].clone())
            used.update(got)
            if not got:
                missing.extend(self.ref_names(s))
        unexpected = [k for k in sd if k not in used]
        if strict and (missing or unexpected):
            raise RuntimeError(f'load_state_dict: missing {missing[:5]}... unexpected {unexpected[:5]}...')
        return missing, unexpected

    def trainable_names(self):
        return [s.name for s in self.specs if s.trainable]
