"""IndoorImVoxelNeck (embodiedscan/models/necks/imvoxel_neck.py:8-143) on the es_hip convolution engine.

The dense (B, C, X, Y, Z) volume is a channels-last row matrix (B*X*Y*Z, C); nn.Conv3d(k=3, stride 1/2, pad 1) and the
1x1 stride-2 down-sample are implicit GEMMs through static dense-grid neighbour maps (es_volume_map), i.e. the same bf16
MFMA kernels as the sparse path with every neighbour present; nn.ConvTranspose3d(k=2, s=2) is eight row GEMMs (the
generative-transpose layout) followed by a row permutation into dense order; BatchNorm3d (train mode) + ReLU (+ the
residual add) is the fused norm kernel.  This block is the one MFMA-bound part of the suite (~4 TFLOP forward
at 40x40x16 with 768 -> 1536 -> 3072 channels)."""
import torch
from ... import engine as E
from ... import hip
from ...hip import P, call
from ...registry import MODELS


class VolumeGrid:
    """static neighbour maps of one dense (B, X, Y, Z) grid"""

    def __init__(self, B, X, Y, Z, dev):
        self.B, self.X, self.Y, self.Z, self.dev = B, X, Y, Z, dev
        self.n = B * X * Y * Z
        self.maps = {}

    def conv_map(self, ks, stride, pad):
        key = (ks, stride, pad)
        if key not in self.maps:
            o = lambda d: (d + 2 * pad - ks) // stride + 1
            Xo, Yo, Zo = o(self.X), o(self.Y), o(self.Z)
            n_out, K = self.B * Xo * Yo * Zo, ks ** 3
            nbr = torch.empty((n_out, K), dtype=torch.int32, device=self.dev)
            call('es_volume_map', self.B, self.X, self.Y, self.Z, Xo, Yo, Zo, ks, stride, pad, P(nbr), hip.stream())
            inv = torch.empty((self.n, K), dtype=torch.int32, device=self.dev)
            call('es_inverse_map', P(nbr), n_out, K, self.n, P(inv), hip.stream())
            self.maps[key] = (nbr, inv, n_out, (Xo, Yo, Zo))
        m = self.maps[key]
        if hip.PROFILE is not None and m[0].data_ptr() not in hip.PAIRS:
            hip.register_map(m[0])
            hip.register_map(m[1])
        return m

    def up_index(self):
        if 'up' not in self.maps:
            idx = torch.empty(self.n * 8, dtype=torch.int32, device=self.dev)
            call('es_volume_up_index', self.B, self.X, self.Y, self.Z, P(idx), hip.stream())
            self.maps['up'] = idx
        return self.maps['up']


class _BN3:
    """nn.BatchNorm3d over the rows of a channels-last volume (train: batch statistics, eval: running statistics)"""

    def __init__(self, arena, prefix):
        g = arena.g
        self.w = E.Param(arena.p[prefix + '.weight'], g.get(prefix + '.weight'))
        self.b = E.Param(arena.p[prefix + '.bias'], g.get(prefix + '.bias'))
        self.running = (arena.p[prefix + '.running_mean'], arena.p[prefix + '.running_var'])

    def __call__(self, x, act=0, res=None, training=True):
        n, C = x.d.shape
        if training:
            return E.norm(x, self.w, self.b, [0, n], 1e-5, act=act, res=res, running=self.running)
        sc = torch.empty(C, dtype=torch.float32, device=x.d.device)
        sh = torch.empty(C, dtype=torch.float32, device=x.d.device)
        call('es_bn_fold', P(self.w.d), P(self.b.d), P(self.running[0]), P(self.running[1]), C, 1e-5, P(sc), P(sh), hip.stream())
        y = E.Var(torch.empty_like(x.d), rg=False)
        call('es_affine_act_fwd', P(x.d), P(sc), P(sh), P(res.d) if res is not None else 0, n, C, act, P(y.d), hip.stream())
        return y


@MODELS.register_module()
class IndoorImVoxelNeck:
    def __init__(self, in_channels, out_channels, n_blocks):
        self.in_channels, self.out_channels, self.n_blocks = in_channels, out_channels, list(n_blocks)
        self.n_scales = len(n_blocks)
        self.training = True
        self.grids = {}

    def bind(self, arena, prefix='neck_3d.'):
        par = lambda n: E.Param(arena.p[prefix + n], arena.g.get(prefix + n))
        bn = lambda n: _BN3(arena, prefix + n)
        self.down, self.up, self.outb = [], {}, []
        c = self.in_channels
        for i, nb in enumerate(self.n_blocks):
            stride = 1 if i == 0 else 2
            layer = []
            for b in range(nb):
                p = f'down_layer_{i}.{b}'
                s = stride if b == 0 else 1
                blk = dict(stride=s, conv1=par(p + '.conv1.weight'), norm1=bn(p + '.norm1'), conv2=par(p + '.conv2.weight'),
                           norm2=bn(p + '.norm2'))
                if s != 1:
                    blk['down'] = (par(p + '.downsample.0.weight'), bn(p + '.downsample.1'))
                    c *= 2
                layer.append(blk)
            self.down.append(layer)
            if i > 0:
                p = f'up_block_{i}'
                self.up[i] = (par(p + '.0.weight'), bn(p + '.1'), par(p + '.3.weight'), bn(p + '.4'))
            p = f'out_block_{i}'
            self.outb.append((par(p + '.0.weight'), bn(p + '.1')))
        return self

    def _grid(self, B, dims, dev):
        key = (B,) + tuple(dims)
        if key not in self.grids:
            self.grids[key] = VolumeGrid(B, dims[0], dims[1], dims[2], dev)
        return self.grids[key]

    def _conv3(self, x, w, g, stride=1):
        """nn.Conv3d(k=3, stride, pad=1): the dense engine (address arithmetic, csrc/dconv.hip) where it takes the shape, the
        neighbour-map kernels elsewhere (the maps are only built if a launch needs them)"""
        o = lambda d: (d + 2 - 3) // stride + 1
        n_out = g.B * o(g.X) * o(g.Y) * o(g.Z)
        return E.conv(x, w, None, None, n_out, dense=(g.B, g.X, g.Y, g.Z, 3, stride, 1),
                      maps=lambda: g.conv_map(3, stride, 1)[:2])

    def _res(self, x, blk, g, B):
        """ResModule (imvoxel_neck.py:112-143)"""
        tr = self.training
        if blk['stride'] == 1:
            o = self._conv3(x, blk['conv1'], g)
            g_out, idt = g, x
        else:
            st = blk['stride']
            dims = tuple((d + 2 - 3) // st + 1 for d in (g.X, g.Y, g.Z))
            o = self._conv3(x, blk['conv1'], g, stride=st)
            g_out = self._grid(B, dims, x.d.device)
            # 1x1x1 stride-2 down-sample of the identity: dense engine (one tap; its data gradient touches class 0 of the parity
            # classes only), the strided maps where a launch falls back
            idt = E.conv(x, blk['down'][0], None, None, B * dims[0] * dims[1] * dims[2], dense=(g.B, g.X, g.Y, g.Z, 1, st, 0),
                         maps=lambda g=g, st=st: g.conv_map(1, st, 0)[:2])
            idt = blk['down'][1](idt, act=0, training=tr)
        o = blk['norm1'](o, act=1, training=tr)
        o = self._conv3(o, blk['conv2'], g_out)
        return blk['norm2'](o, act=1, res=idt, training=tr), g_out          # relu(bn(conv2) + identity)

    def forward(self, x, dims, B=1, on_coarse=None):
        """x: Var (B*X*Y*Z, C_in) channels-last, dims = (X, Y, Z).  Returns [(Var (B*Xi*Yi*Zi, out_channels), (Xi,Yi,Zi))]
        fine -> coarse (imvoxel_neck.py:34-58)."""
        tr = self.training
        g = self._grid(B, dims, x.d.device)
        down = []
        for li, layer in enumerate(self.down):
            if on_coarse is not None and li == self.n_scales - 1:
                on_coarse()                     # tape position: everything recorded from here on is the coarse half of the neck
            for blk in layer:
                x, g = self._res(x, blk, g, B)
            down.append((x, g))
        outs = []
        for i in range(self.n_scales - 1, -1, -1):
            if i < self.n_scales - 1:
                wt, bn1, wc, bn2 = self.up[i + 1]
                gf = down[i][1]
                geo = (g.B, g.X, g.Y, g.Z, 2, 2, 0)
                ci_t, co_t = wt.d.shape[1], wt.d.shape[2]
                if x.d.stride(0) == ci_t and all(E.dense_ok(geo, m, ci_t, co_t) for m in (3, 4, 5)):
                    up = E.conv_transpose_dense(x, wt, geo)                       # dense engine: written straight into dense order
                else:
                    up = E.gather_rows(E.gen_conv_transpose(x, wt), g.up_index())     # rows 8*i+tap -> dense order
                assert up.d.shape[0] == gf.n, 'odd volume sizes are not supported by ConvTranspose3d(k=2,s=2) + add'
                up = bn1(up, act=1, training=tr)
                up = bn2(self._conv3(up, wc, gf), act=1, training=tr)
                x, g = E.add(down[i][0], up), gf
            wo, bno = self.outb[i]
            outs.append((bno(self._conv3(x, wo, g), act=1, training=tr), (g.X, g.Y, g.Z)))
        return outs[::-1]

    __call__ = forward
