"""MinkNeck (embodiedscan/models/necks/mink_neck.py:25-244) on the MI355X kernels: the FCAF3D-style sparse top-down
pyramid of the grounder -- generative transposed conv + BN + ELU + 3^3 conv + BN + ELU, sparse union add, score-based
pruning to `pts_prune_threshold` voxels per sample (1000 in the shipped config, so the pruning path is always live),
3^3 output conv to 256 channels, and the 1x1 `conv_cls` whose max only ranks voxels for the next pruning step.
Returns what the reference returns (per-sample lists, levels concatenated coarse -> fine) PLUS the padded batch layout
the transformer kernels consume, so that no per-sample Python concatenation of device tensors is needed."""
import torch
from ... import engine as E
from ... import hip
from ... import sparse
from ...hip import P, call, iarr
from ...registry import MODELS
from ...sparse import SparseTensor
from ..dense_heads.fcaf3d_head import _BN, conv3


@MODELS.register_module()
class MinkNeck:
    def __init__(self, num_classes, in_channels, out_channels, voxel_size, pts_prune_threshold, train_cfg=None,
                 test_cfg=None, init_cfg=None):
        self.num_classes, self.in_channels, self.out_channels = num_classes, tuple(in_channels), out_channels
        self.voxel_size, self.pts_prune_threshold = voxel_size, pts_prune_threshold
        self.training = True

    def bind(self, arena, prefix='neck_3d.'):
        par = lambda n: E.Param(arena.p[prefix + n], arena.g.get(prefix + n))
        self.up, self.out = {}, {}
        for i in range(len(self.in_channels)):
            if i > 0:
                p = f'up_block_{i}'
                self.up[i] = (par(p + '.0.kernel'), _BN(arena, prefix + p + '.1'), par(p + '.3.kernel'), _BN(arena, prefix + p + '.4'))
            p = f'out_block_{i}'
            self.out[i] = (par(p + '.0.kernel'), _BN(arena, prefix + p + '.1'))
        self.cls_w, self.cls_b = arena.p[prefix + 'conv_cls.kernel'], arena.p[prefix + 'conv_cls.bias']
        return self

    def _prune(self, x, score_set, score):
        """mink_neck.py:178-204"""
        off = x.cs.offsets()
        thr = self.pts_prune_threshold
        if all(off[b + 1] - off[b] <= thr for b in range(x.cs.n_batch)):
            return x
        idx, w = sparse.interp_map(x.cs, score_set)
        s = torch.empty(x.cs.n, dtype=torch.float32, device=x.cs.device)
        call('es_interp_scores', P(score), P(idx), P(w), x.cs.n, P(s), hip.stream())
        mask = torch.empty(x.cs.n, dtype=torch.int32, device=x.cs.device)
        ws, nws = E.ticket_ws(int(hip.raw('es_topk_mask_workspace_ints')(x.cs.n_batch)), s, tag='topk')
        call('es_topk_mask_ws', P(s), iarr(off), x.cs.n_batch, int(thr), P(mask), P(ws), nws, hip.stream())
        kept = [0]                                  # top-k keeps min(n_b, thr) rows of sample b: no row-count read-back
        for b in range(x.cs.n_batch):
            kept.append(kept[-1] + min(off[b + 1] - off[b], thr))
        new_set, src = sparse.compact(x.cs, mask, offsets=kept)
        return SparseTensor(new_set, E.gather_rows(x.F, src))

    def prefetch_coords(self, level_sets, maps=False):
        """feature-independent coordinate work of the top-down pass ahead of the feature kernels (see
        FCAF3DHeadRotMat.prefetch_coords); stops at the first level whose pruning is live"""
        x = level_sets[-1]
        thr = self.pts_prune_threshold
        if maps:
            x.kernel_map(x, 3), x.inverse_map(x, 3)
        for i in range(len(level_sets) - 2, -1, -1):
            c = x.children()
            if maps:
                c.kernel_map(c, 3), c.inverse_map(c, 3)
            u, _, _ = sparse.union(c, level_sets[i])          # (generated children first: see _levels)
            off = u.offsets()
            if any(off[b + 1] - off[b] > thr for b in range(u.n_batch)):
                break
            if maps:
                u.kernel_map(u, 3), u.inverse_map(u, 3)
            x = u

    def levels(self, inputs):
        """-> per level i (input order: fine .. coarse) dict(cs, out Var (n, out_channels), cls (n, num_classes))"""
        n_lvl = len(inputs)
        res = [None] * n_lvl
        x = inputs[-1]
        score_set = score = None
        tr = self.training
        for i in range(n_lvl - 1, -1, -1):
            if i < n_lvl - 1:
                wt, bn1, wc, bn2 = self.up[i + 1]
                y = SparseTensor(x.cs.children(), bn1(E.gen_conv_transpose(x.F, wt), act=2, training=tr))
                y = conv3(y, wc)
                y = SparseTensor(y.cs, bn2(y.F, act=2, training=tr))
                # union rows = the generated children (Z order, inherited from the parents) followed by the few backbone voxels
                # they do not cover -- NOT backbone first (rounds 1-5): a + b is the same sum, but with the children appended behind
                # the backbone rows a 256-row tile's 3x3x3 neighbourhood straddled two row ranges and its halo (csrc/halo.hip) grew
                # from ~470 to ~600-700 source rows (profiles/r6d_halo_stats_*.txt); ME's own row order is hash order, ours is a spec
                # the oracle shares (oracle/model.py, oracle/grounding.py)
                u, pa, pb = sparse.union(y.cs, inputs[i].cs)
                x = SparseTensor(u, E.union_add(y.F, inputs[i].F, pa, pb, u.n))
                x = self._prune(x, score_set, score)
            wo, bno = self.out[i]
            out = conv3(x, wo)
            out = SparseTensor(out.cs, bno(out.F, act=2, training=tr))
            n = out.cs.n
            # conv_cls (1x1, bias): ranking scores only, no gradient (used under no_grad by _prune) -> exact-f32 row GEMM
            cls = torch.empty((n, self.num_classes), dtype=torch.float32, device=out.F.d.device)
            call('es_spconv_fwd', P(out.F.d), out.F.d.stride(0), P(self.cls_w), 0, n, n, 1, self.out_channels, self.num_classes,
                 P(self.cls_b), P(cls), self.num_classes, 0, 0, hip.stream())
            score = torch.empty(n, dtype=torch.float32, device=cls.device)
            call('es_row_max', P(cls), self.num_classes, n, self.num_classes, P(score), hip.stream())
            score_set = out.cs
            res[i] = dict(cs=out.cs, out=out.F, cls=cls)
        return res

    def forward(self, x, batch_size):
        """mink_neck.py:133-176, 206-244: (batch_feats_list, batch_scores_list, batch_points_list): per sample, levels
        concatenated coarse -> fine.  The lists are row slices of the padded buffers kept in `self.last`:
        feats Var (B*Lmax, C) (zero rows behind each sample's `lens[b]` rows), points (B*Lmax, 3), lens, Lmax."""
        lv = self.levels(x)
        B = batch_size
        order = list(range(len(lv) - 1, -1, -1))               # coarse -> fine, as the reference appends them
        offs = {i: lv[i]['cs'].offsets() for i in order}
        lens = [sum(offs[i][b + 1] - offs[i][b] for i in order) for b in range(B)]
        Lmax = max(lens)
        C = self.out_channels
        dev = lv[0]['out'].d.device
        feats = torch.zeros((B * Lmax, C), dtype=torch.float32, device=dev)
        pts = torch.zeros((B * Lmax, 3), dtype=torch.float32, device=dev)
        scores = torch.zeros((B * Lmax, self.num_classes), dtype=torch.float32, device=dev)
        maps = []
        pos = [0] * B
        s = hip.stream()
        for i in order:
            n = lv[i]['cs'].n
            dst = torch.empty(n, dtype=torch.int32)
            for b in range(B):
                r0, r1 = offs[i][b], offs[i][b + 1]
                dst[r0:r1] = torch.arange(b * Lmax + pos[b], b * Lmax + pos[b] + (r1 - r0), dtype=torch.int32)
                pos[b] += r1 - r0
            dst = dst.to(dev, non_blocking=True)
            p = torch.empty((n, 3), dtype=torch.float32, device=dev)
            call('es_coords_to_points', P(lv[i]['cs'].coords), n, float(self.voxel_size), P(p), s)
            call('es_row_move', P(feats), C, P(lv[i]['out'].d), C, P(dst), n, C, 2, s)
            call('es_row_move', P(pts), 3, P(p), 3, P(dst), n, 3, 2, s)
            call('es_row_move', P(scores), self.num_classes, P(lv[i]['cls']), self.num_classes, P(dst), n, self.num_classes, 2, s)
            maps.append((lv[i]['out'], dst, n))
        fv = E.Var(feats)

        def bwd():
            if fv.g is None:
                return
            for out, dst, n in maps:
                g = torch.empty_like(out.d)
                call('es_row_move', P(g), C, P(fv.g), C, P(dst), n, C, 0, hip.stream())
                if out.g is None:
                    out.g = g
                else:
                    E.add_into(out.g, g)
        E.TAPE.add(bwd)
        self.last = dict(feats=fv, points=pts, scores=scores, lens=lens, Lmax=Lmax, levels=lv)
        fl = [feats[b * Lmax:b * Lmax + lens[b]] for b in range(B)]
        sl = [scores[b * Lmax:b * Lmax + lens[b]] for b in range(B)]
        pl = [pts[b * Lmax:b * Lmax + lens[b]] for b in range(B)]
        return fl, sl, pl

    __call__ = forward
