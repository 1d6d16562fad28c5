"""MinkResNet (depth 34, ME BasicBlocks) on the es_hip sparse convolution engine.
Same constructor as embodiedscan/models/backbones/mink_resnet.py:20-140; forward takes / returns
embodiedscan_amd.sparse.SparseTensor instead of ME.SparseTensor."""
from ... import engine as E
from ...registry import MODELS
from ...sparse import SparseTensor
from ..dense_heads.fcaf3d_head import _BN


@MODELS.register_module()
class MinkResNet:
    arch_settings = {18: (2, 2, 2, 2), 34: (3, 4, 6, 3)}

    def __init__(self, depth, in_channels, num_stages=4, pool=True):
        if depth not in self.arch_settings:
            raise KeyError(f'invalid depth {depth} for the BasicBlock MinkResNet implemented here')
        assert 4 >= num_stages >= 1
        self.depth, self.in_channels, self.num_stages, self.pool = depth, in_channels, num_stages, pool
        self.stage_blocks = self.arch_settings[depth][:num_stages]
        self.training = True
        self.trace = None          # debugging aid: list collecting every block output Var

    def bind(self, arena, prefix='backbone_3d.'):
        self.arena, self.prefix = arena, prefix
        par = lambda n: E.Param(arena.p[prefix + n], arena.g.get(prefix + n))
        self.conv1 = par('conv1.kernel')
        self.norm1 = (par('norm1.weight'), par('norm1.bias'))
        self.blocks = []
        for li, nblk in enumerate(self.stage_blocks):
            layer = []
            for bi in range(nblk):
                p = f'layer{li + 1}.{bi}.'
                blk = dict(conv1=par(p + 'conv1.kernel'), norm1=_BN(arena, prefix + p + 'norm1'),
                           conv2=par(p + 'conv2.kernel'), norm2=_BN(arena, prefix + p + 'norm2'))
                if bi == 0:
                    blk['down'] = (par(p + 'downsample.0.kernel'), _BN(arena, prefix + p + 'downsample.1'))
                layer.append(blk)
            self.blocks.append(layer)
        return self

    def prefetch_coords(self, cs):
        """every strided coordinate set of the forward pass (each costs one row-count read-back) ahead of the feature
        kernels; returns the sets of the output levels.  The results are cached on the sets, forward() re-uses them."""
        # all levels and their per-sample offsets in one host round trip (sparse.strided_chain; ES_COORD_BATCH=0: the chain of
        # one read-back per level and per offsets vector that rounds 1-3 used)
        from ... import sparse as _sp
        sets = _sp.strided_chain(cs, 1 + int(bool(self.pool)) + len(self.blocks))
        sets[0].offsets()                  # per-sample segments of the instance norm behind conv1 (already on the host)
        outs = sets[1 + int(bool(self.pool)):]
        for o in outs:
            o.offsets()
        return outs

    def prefetch_maps(self, cs):
        """build every kernel / inverse map forward() and its backward will ask for (cached on the sets; no host round trip):
        the detector's next-batch prefetch calls this under the previous step's backward pass"""
        o1 = cs.strided(2)
        cs.kernel_map(o1, 3)
        # (conv1's data gradient is never taken: its input is the raw point feature)
        if self.pool:
            o2 = o1.strided(2)
            o1.kernel_map(o2, 2)
            cs = o2
        else:
            cs = o1
        for layer in self.blocks:
            oc = cs.strided(2)
            for k in (3, 1):
                cs.kernel_map(oc, k)
                cs.inverse_map(oc, k)
            cs = oc
            cs.kernel_map(cs, 3)                         # conv2 of every block, conv1 of the blocks behind the first
            cs.inverse_map(cs, 3)

    def forward(self, x):
        tr = self.training
        cs = x.cs
        # conv1: k3 s2, instance norm, ReLU, max pool k2 s2   (mink_resnet.py:131-135)
        o1 = cs.strided(2)
        f = E.conv(x.F, self.conv1, cs.kernel_map(o1, 3), cs.inverse_map(o1, 3), o1.n, need_dx=x.F.rg)
        f = E.norm(f, self.norm1[0], self.norm1[1], o1.offsets(), 1e-8, act=1)
        if self.pool:
            o2 = o1.strided(2)
            f = E.maxpool(f, o1.kernel_map(o2, 2), o2.n)
            cs = o2
        else:
            cs = o1
        outs = []
        for layer in self.blocks:
            for bi, blk in enumerate(layer):
                if bi == 0:
                    oc = cs.strided(2)
                    o = E.conv(f, blk['conv1'], cs.kernel_map(oc, 3), cs.inverse_map(oc, 3), oc.n)
                    idt = E.conv(f, blk['down'][0], cs.kernel_map(oc, 1), cs.inverse_map(oc, 1), oc.n)
                    idt = blk['down'][1](idt, act=0, training=tr)
                    cs = oc
                else:
                    o = E.conv(f, blk['conv1'], cs.kernel_map(cs, 3), cs.inverse_map(cs, 3), cs.n)
                    idt = f
                c1 = o
                o = blk['norm1'](o, act=1, training=tr)
                n1 = o
                o = E.conv(o, blk['conv2'], cs.kernel_map(cs, 3), cs.inverse_map(cs, 3), cs.n)
                f = blk['norm2'](o, act=1, res=idt, training=tr)       # relu(bn(conv2) + identity)
                if self.trace is not None:
                    self.trace += [c1, n1, o, f]
            outs.append(SparseTensor(cs, f))
        return outs

    __call__ = forward
