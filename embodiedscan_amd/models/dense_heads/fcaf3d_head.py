"""FCAF3DHeadRotMat on the MI355X kernels.

Same constructor arguments, loss keys and call protocol as the reference class
(embodiedscan/models/dense_heads/fcaf3d_head.py:827-1725); the sparse FPN + head run on the
es_hip convolution engine, target assignment / focal / box-coder + corner-Chamfer losses are
single fused kernels that also emit the gradients of the head outputs (no autograd graph, no
per-sample host sync inside the loss).
"""
import torch
from ... import hip
from ... import engine as E
from ... import sparse
from ...geometry import euler_to_matrix_zxy
from ...hip import P, call, iarr, farr, parr
from ...parallel import reduce_mean
from ...registry import MODELS
from ...sparse import SparseTensor


def _stream():
    return hip.stream()


def upload_gts(gts, dev):
    """Ground truth of a whole batch -> device in ONE copy from pinned memory (issued before the head runs, so the
    loss section never waits on a host-to-device transfer).  gts: [(boxes (G,9), labels (G,))] per sample.
    Returns per sample (boxes (G,9) f32, R(-euler) (G,9) f32, labels (G,) int32) device views."""
    Gs = [int(b.shape[0]) for b, _ in gts]
    tot = sum(Gs)
    host = torch.empty(max(tot, 1) * 19, dtype=torch.float32, pin_memory=dev.type == 'cuda')
    hb, hr, hl = host[:tot * 9].view(tot, 9), host[tot * 9:tot * 18].view(tot, 9), host[tot * 18:tot * 19].view(torch.int32)
    a = 0
    for (b, l), G in zip(gts, Gs):
        if G:
            bh = b.detach().float().cpu()
            hb[a:a + G] = bh
            hr[a:a + G] = euler_to_matrix_zxy(-bh[:, 6:9]).reshape(G, 9)
            hl[a:a + G] = l.detach().to(torch.int32).cpu()
        a += G
    d = host.to(dev, non_blocking=True)
    db, dr, dl = d[:tot * 9].view(tot, 9), d[tot * 9:tot * 18].view(tot, 9), d[tot * 18:tot * 19].view(torch.int32)
    out, a = [], 0
    for G in Gs:
        out.append((db[a:a + G], dr[a:a + G], dl[a:a + G]))
        a += G
    return out


def get_targets_device(points_per_level, gt_boxes, gt_labels, assign_thr, center_thr, dev_gt=None):
    """A12 for one sample.  points_per_level: device (n_l,3) f32 tensors (or one concatenated tensor +
    level offsets as a tuple).  gt_boxes (G,9) / gt_labels (G,) may live on the host; dev_gt: the sample's entry of
    upload_gts() when the batch was uploaded ahead of time.
    Returns center_t (N,), bbox_t (N,9), cls_t (N,) int32, box_idx (N,) int32, n_pos (1,) int32 (device)."""
    if isinstance(points_per_level, tuple):
        points, level_off = points_per_level
    else:
        points = torch.cat(points_per_level)
        level_off = [0]
        for p in points_per_level:
            level_off.append(level_off[-1] + int(p.shape[0]))
    dev = points.device
    N, G = int(points.shape[0]), int(gt_boxes.shape[0])
    if dev_gt is not None:
        boxes, rot, labels = dev_gt
    else:
        boxes_h = gt_boxes.detach().float().cpu()
        rot_neg = euler_to_matrix_zxy(-boxes_h[:, 6:9]).reshape(G, 9) if G else torch.zeros((0, 9))
        boxes = boxes_h.to(dev).contiguous()
        rot = rot_neg.to(dev).contiguous()
        labels = gt_labels.detach().to(torch.int32).to(dev).contiguous()
    n_lvl = len(level_off) - 1
    scratch = torch.empty(max(G, 1) * max(N, 1) + (n_lvl + 2) * max(G, 1) + 8, dtype=torch.float32, device=dev)
    center_t = torch.empty(N, dtype=torch.float32, device=dev)
    bbox_t = torch.empty((N, 9), dtype=torch.float32, device=dev)
    cls_t = torch.empty(N, dtype=torch.int32, device=dev)
    box_idx = torch.empty(N, dtype=torch.int32, device=dev)
    n_pos = torch.empty(1, dtype=torch.int32, device=dev)
    call('es_get_targets', P(points), N, iarr(level_off), n_lvl, P(boxes), P(rot), P(labels), G, int(assign_thr),
         int(center_thr), P(scratch), P(center_t), P(bbox_t), P(cls_t), P(box_idx), P(n_pos), _stream())
    return center_t, bbox_t, cls_t, box_idx, n_pos


class _BN:
    def __init__(self, arena, prefix):
        g = arena.g
        self.w = E.Param(arena.p[prefix + '.bn.weight'], g.get(prefix + '.bn.weight'))
        self.b = E.Param(arena.p[prefix + '.bn.bias'], g.get(prefix + '.bn.bias'))
        self.running = (arena.p[prefix + '.bn.running_mean'], arena.p[prefix + '.bn.running_var'])

    def __call__(self, x, act=0, res=None, training=True):
        n = x.d.shape[0]
        if training:
            return E.norm(x, self.w, self.b, [0, n], 1e-5, act=act, res=res, running=self.running)
        # eval mode (nn.BatchNorm1d.eval()): running statistics folded to scale / shift, forward only
        C = self.w.d.numel()
        sc, sh = torch.empty(C, dtype=torch.float32, device=x.d.device), torch.empty(C, dtype=torch.float32, device=x.d.device)
        call('es_bn_fold', P(self.w.d), P(self.b.d), P(self.running[0]), P(self.running[1]), C, 1e-5, P(sc), P(sh), _stream())
        y = E.Var(torch.empty_like(x.d), rg=False)
        call('es_affine_act_fwd', P(x.d), P(sc), P(sh), P(res.d) if res is not None else 0, n, C, act, P(y.d), _stream())
        return y


def conv3(st, w, out_set=None):
    """3^3 MinkowskiConvolution on the same coordinate set."""
    cs = st.cs
    nbr, inv = cs.kernel_map(cs, 3), cs.inverse_map(cs, 3)
    return SparseTensor(cs, E.conv(st.F, w, nbr, inv, cs.n))


@MODELS.register_module()
class FCAF3DHeadRotMat:
    def __init__(self, num_classes, in_channels, out_channels, num_reg_outs, voxel_size, pts_prune_threshold,
                 pts_assign_threshold, pts_center_threshold, center_loss=None, bbox_loss=None, cls_loss=None,
                 decouple_bbox_loss=False, decouple_groups=3, decouple_weights=None, norm_decouple_loss=False,
                 train_cfg=None, test_cfg=None, init_cfg=None):
        self.num_classes, self.in_channels, self.out_channels = num_classes, tuple(in_channels), out_channels
        self.num_reg_outs = num_reg_outs
        self.voxel_size = voxel_size
        self.pts_prune_threshold = pts_prune_threshold
        self.pts_assign_threshold = pts_assign_threshold
        self.pts_center_threshold = pts_center_threshold
        center_loss = center_loss or dict(type='mmdet.CrossEntropyLoss', use_sigmoid=True)
        bbox_loss = bbox_loss or dict(type='BBoxCDLoss', mode='l1', loss_weight=1.0, group='g8')
        cls_loss = cls_loss or dict(type='mmdet.FocalLoss')
        # the fused loss kernels implement exactly the shipped configuration
        assert center_loss.get('type') == 'mmdet.CrossEntropyLoss' and center_loss.get('use_sigmoid', False)
        assert bbox_loss.get('type') == 'BBoxCDLoss' and bbox_loss.get('mode', 'l2') == 'l1' and \
            bbox_loss.get('group', 'g8') == 'g8'
        assert cls_loss.get('type') == 'mmdet.FocalLoss'
        assert num_reg_outs == 12, 'only the 6D-rotation (12 output) coder is implemented'
        self.bbox_loss_weight = float(bbox_loss.get('loss_weight', 1.0))
        self.focal_gamma, self.focal_alpha = float(cls_loss.get('gamma', 2.0)), float(cls_loss.get('alpha', 0.25))
        self.decouple_bbox_loss = decouple_bbox_loss
        self.decouple_groups = decouple_groups
        assert not norm_decouple_loss, 'norm_decouple_loss is not used by the shipped configs'
        if decouple_weights is None:
            decouple_weights = [1.0 / decouple_groups] * decouple_groups
        self.decouple_weights = list(decouple_weights)
        self.train_cfg, self.test_cfg = train_cfg, test_cfg
        self.training = True

    # ------------------------------------------------------------------ parameters
    def bind(self, arena, prefix='bbox_head.'):
        g = arena.g
        self.arena, self.prefix = arena, prefix
        par = lambda n: E.Param(arena.p[prefix + n], g.get(prefix + n))
        n_lvl = len(self.in_channels)
        self.up, self.out = {}, {}
        for i in range(n_lvl):
            if i > 0:
                p = f'up_block_{i}'
                self.up[i] = (par(p + '.0.kernel'), _BN(arena, prefix + p + '.1'), par(p + '.3.kernel'),
                              _BN(arena, prefix + p + '.4'))
            p = f'out_block_{i}'
            self.out[i] = (par(p + '.0.kernel'), _BN(arena, prefix + p + '.1'))
        self.head_w = par('head_out.kernel')             # (1, out_channels, 1 + 12 + num_classes padded to a multiple of 64)
        self.head_b = par('head_out.bias')               # (1 + 12 + num_classes,), only the class part is live
        self.scales = [par(f'scales.{i}.scale') for i in range(n_lvl)]
        return self

    # ------------------------------------------------------------------ forward
    def _prune(self, x, score_set, score):
        """fcaf3d_head.py:1091-1114.  Fast path: every sample already has <= threshold rows."""
        off = x.cs.offsets()
        thr = self.pts_prune_threshold
        if all(off[b + 1] - off[b] <= thr for b in range(x.cs.n_batch)):
            return x
        idx, w = sparse.interp_map(x.cs, score_set)
        s = torch.empty(x.cs.n, dtype=torch.float32, device=x.cs.device)
        call('es_interp_scores', P(score), P(idx), P(w), x.cs.n, P(s), _stream())
        mask = torch.empty(x.cs.n, dtype=torch.int32, device=x.cs.device)
        ws, nws = E.ticket_ws(int(hip.raw('es_topk_mask_workspace_ints')(x.cs.n_batch)), s, tag='topk')
        call('es_topk_mask_ws', P(s), iarr(off), x.cs.n_batch, int(thr), P(mask), P(ws), nws, _stream())
        kept = [0]                                  # top-k keeps min(n_b, thr) rows of sample b: no row-count read-back
        for b in range(x.cs.n_batch):
            kept.append(kept[-1] + min(off[b + 1] - off[b], thr))
        new_set, src = sparse.compact(x.cs, mask, offsets=kept)
        return SparseTensor(new_set, E.gather_rows(x.F, src))

    def prefetch_coords(self, level_sets, maps=False):
        """Coordinate work of the top-down pass that does not depend on features -- generative children, unions and their
        per-sample offsets (each a small device->host read-back) -- done BEFORE the feature kernels are queued, while the
        image branch keeps the GPU busy.  The forward pass then finds everything cached and issues without host stalls.
        Stops at the first level whose pruning is live (its output set depends on scores)."""
        x = level_sets[-1]
        thr = self.pts_prune_threshold
        if maps:                                        # next-batch prefetch: the 3^3 maps of the out_block convolutions too
            x.kernel_map(x, 3), x.inverse_map(x, 3)
        for i in range(len(level_sets) - 2, -1, -1):
            c = x.children()
            if maps:                                    # up_block's 3^3 convolution runs on the generated children
                c.kernel_map(c, 3), c.inverse_map(c, 3)
            u, _, _ = sparse.union(c, level_sets[i])          # (generated children first: see _levels)
            off = u.offsets()
            if any(off[b + 1] - off[b] > thr for b in range(u.n_batch)):
                break                                   # pruning is live: the set the out_block sees depends on scores
            if maps:
                u.kernel_map(u, 3), u.inverse_map(u, 3)
            x = u

    def _levels(self, inputs):
        """Runs the sparse FPN + head.  Returns per level (fine->coarse) a dict with the
        coordinate set, the raw head GEMM output Var (n, 1+12+C) and the decoded boxes (n,12)."""
        n_lvl = len(inputs)
        levels = [None] * n_lvl
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
            ho = E.conv(out.F, self.head_w, None, None, n, bias=self.head_b, bias_from=13)
            ncol = ho.d.shape[1]
            bbox = torch.empty((n, 12), dtype=torch.float32, device=ho.d.device)
            call('es_reg_decode_fwd', ho.d.data_ptr() + 4, ncol, n, P(self.scales[i].d), P(bbox), _stream())
            score = torch.empty(n, dtype=torch.float32, device=ho.d.device)
            call('es_row_max', ho.d.data_ptr() + 4 * 13, ncol, n, self.num_classes, P(score), _stream())
            score_set = out.cs
            levels[i] = dict(cs=out.cs, ho=ho, bbox=bbox, scale=self.scales[i], out=out.F, x=x.F)
        return levels

    def forward(self, x):
        """Reference return format (fcaf3d_head.py:993-1020): four lists over levels (fine->coarse) of lists
        over samples: center (n,1), bbox (n,12), cls (n,C), points (n,3)."""
        levels = self._levels(x)
        cp, bp, kp, pts = [], [], [], []
        for lv in levels:
            off = lv['cs'].offsets()
            ho, bbox = lv['ho'].d, lv['bbox']
            points = torch.empty((lv['cs'].n, 3), dtype=torch.float32, device=ho.device)
            call('es_coords_to_points', P(lv['cs'].coords), lv['cs'].n, float(self.voxel_size), P(points), _stream())
            sl = [slice(off[b], off[b + 1]) for b in range(lv['cs'].n_batch)]
            cp.append([ho[s, 0:1] for s in sl])
            bp.append([bbox[s] for s in sl])
            kp.append([ho[s, 13:13 + self.num_classes] for s in sl])
            pts.append([points[s] for s in sl])
        return cp, bp, kp, pts

    __call__ = forward

    # ------------------------------------------------------------------ predict
    def predict(self, x, batch_data_samples, rescale=False):
        """fcaf3d_head.py:1052-1089,1352-1431: list of InstanceData(bboxes_3d, scores_3d, labels_3d)."""
        from ...structures import EulerDepthInstance3DBoxes, InstanceData
        cfg = self.test_cfg or {}
        nms_pre, score_thr, iou_thr = cfg.get('nms_pre', 1000), cfg.get('score_thr', 0.01), cfg.get('iou_thr', 0.5)
        prev = E.TAPE.enabled
        E.TAPE.enabled = False
        try:
            levels = self._levels(x)
        finally:
            E.TAPE.enabled = prev
        dev = levels[0]['ho'].d.device
        s = _stream()
        B = levels[0]['cs'].n_batch
        C = self.num_classes
        per_level = []
        for lv in levels:
            n = lv['cs'].n
            ho = lv['ho'].d
            scores = torch.empty((n, C), dtype=torch.float32, device=dev)
            maxs = torch.empty(n, dtype=torch.float32, device=dev)
            call('es_predict_scores', P(ho), ho.shape[1], n, C, P(scores), P(maxs), s)
            pts = torch.empty((n, 3), dtype=torch.float32, device=dev)
            call('es_coords_to_points', P(lv['cs'].coords), n, float(self.voxel_size), P(pts), s)
            off = lv['cs'].offsets()
            mask = torch.empty(n, dtype=torch.int32, device=dev)
            call('es_topk_mask', P(maxs), iarr(off), B, int(nms_pre) if nms_pre > 0 else n + 1, P(mask), s)
            kept_set, src = sparse.compact(lv['cs'], mask)           # rows kept by the per-sample top-nms_pre
            boxes = torch.empty((kept_set.n, 9), dtype=torch.float32, device=dev)
            call('es_decode_boxes', P(pts), P(lv['bbox']), P(src), kept_set.n, P(boxes), s)
            sc_k = torch.empty((kept_set.n, C), dtype=torch.float32, device=dev)
            call('es_row_move', P(sc_k), C, P(scores), C, P(src), kept_set.n, C, 0, s)
            per_level.append((kept_set.offsets(), boxes, sc_k))
        results = []
        for b in range(B):
            bx = torch.cat([bl[off[b]:off[b + 1]] for off, bl, _ in per_level])
            sc = torch.cat([sl[off[b]:off[b + 1]] for off, _, sl in per_level])
            M = int(bx.shape[0])
            keep_idx = torch.empty((C, max(M, 1)), dtype=torch.int32, device=dev)
            keep_cnt = torch.zeros(C, dtype=torch.int32, device=dev)
            if M:
                call('es_nms3d_multiclass', P(bx), P(sc), M, C, float(score_thr), float(iou_thr), P(keep_idx), P(keep_cnt), s)
            cnt = keep_cnt.cpu().tolist()
            ob, os_, ol = [], [], []
            for c in range(C):
                if cnt[c]:
                    ids = keep_idx[c, :cnt[c]].long()
                    ob.append(bx[ids]); os_.append(sc[ids, c])
                    ol.append(torch.full((cnt[c],), c, dtype=torch.long, device=dev))
            if ob:
                rb, rs, rl = torch.cat(ob), torch.cat(os_), torch.cat(ol)
                # Reference quirk, reproduced literally (fcaf3d_head.py:1681-1682 keeps only (x,y,z,dx,dy,dz,alpha)
                # through NMS, then euler_box3d.py:44-48 pads beta = gamma = 0): the emitted boxes are yaw-only.
                rb[:, 7:] = 0
            else:
                rb, rs, rl = bx.new_zeros((0, 9)), bx.new_zeros((0,)), torch.zeros((0,), dtype=torch.long, device=dev)
            results.append(InstanceData(bboxes_3d=EulerDepthInstance3DBoxes(rb), scores_3d=rs, labels_3d=rl))
        return results

    # ------------------------------------------------------------------ loss
    def loss(self, x, batch_data_samples, **kwargs):
        """fcaf3d_head.py:1022-1050.  Returns dict(loss_center, loss_bbox, loss_cls) (device scalars) and
        seeds the gradients of the head outputs on the tape (call engine.TAPE.backward())."""
        gts = [(ds.gt_instances_3d.bboxes_3d, ds.gt_instances_3d.labels_3d) for ds in batch_data_samples]
        gts = [(getattr(b, 'tensor', b), l) for b, l in gts]
        dev_gts = upload_gts(gts, x[0].F.d.device)             # before the head's launches are queued
        levels = self._levels(x)
        return self.loss_by_levels(levels, gts, dev_gts)

    def loss_by_levels(self, levels, gts, dev_gts=None):
        dev = levels[0]['ho'].d.device
        if dev_gts is None:
            dev_gts = upload_gts(gts, dev)
        B = len(gts)
        n_lvl = len(levels)
        s = _stream()
        offs = [lv['cs'].offsets() for lv in levels]
        # locations in metres, per level
        for lv in levels:
            pts = torch.empty((lv['cs'].n, 3), dtype=torch.float32, device=dev)
            call('es_coords_to_points', P(lv['cs'].coords), lv['cs'].n, float(self.voxel_size), P(pts), s)
            lv['points'] = pts
            lv['dho'] = torch.zeros_like(lv['ho'].d)
            lv['dbbox'] = torch.zeros_like(lv['bbox'])
        # phase 1: targets per sample (points of a sample concatenated fine->coarse, fcaf3d_head.py:1595-1600)
        # Samples are independent until the avg_factor below and again after it: odd samples go to the side stream.
        two = E.TWO_STREAMS[0] and B > 1
        on_main = [b for b in range(B) if not (two and b % 2)]
        on_side = [b for b in range(B) if two and b % 2]
        per = [None] * B
        n_pos_all = torch.empty(B, dtype=torch.int32, device=dev)

        def targets(b):
            lo = [0]
            for l in range(n_lvl):
                lo.append(lo[-1] + offs[l][b + 1] - offs[l][b])
            pts = torch.cat([levels[l]['points'][offs[l][b]:offs[l][b + 1]] for l in range(n_lvl)])
            ct, bt, kt, bi, npos = get_targets_device((pts, lo), gts[b][0], gts[b][1], self.pts_assign_threshold,
                                                      self.pts_center_threshold, dev_gt=dev_gts[b])
            n_pos_all[b:b + 1] = npos
            per[b] = (pts, lo, ct, bt, kt, npos)

        if on_side:
            with E.side_stream():
                for b in on_side:
                    targets(b)
        for b in on_main:
            targets(b)
        if on_side:
            E.join_side()
        # phase 2: avg_factor = max(reduce_mean(n_pos), 1) for all samples in ONE collective
        avg = reduce_mean(n_pos_all.float()).clamp(min=1.0).contiguous()
        # phase 3: losses + gradients, per (sample, level) slice, no host sync
        loss_cls = torch.zeros(B, dtype=torch.float32, device=dev)
        loss_acc = torch.zeros((B, 2), dtype=torch.float64, device=dev)       # f64 sums (order-independent after rounding)
        partials = [torch.empty(2048, dtype=torch.float64, device=dev) for _ in range(2)]   # one scratch per stream
        gw = [w * self.bbox_loss_weight for w in self.decouple_weights]
        if not self.decouple_bbox_loss:
            gw = [0., 0., 0., self.bbox_loss_weight]
        elif self.decouple_groups == 3:
            gw = gw[:3] + [0.]
        gwa = farr(gw)
        gscale = 1.0 / B
        def sample_losses(b, partial):
            s = _stream()
            pts, lo, ct, bt, kt, npos = per[b]
            ncol = levels[0]['ho'].d.shape[1]
            hos, bbs, dhos, dbbs = [], [], [], []
            for l in range(n_lvl):
                lv = levels[l]
                r0 = offs[l][b]
                hos.append(lv['ho'].d.data_ptr() + 4 * r0 * ncol)
                dhos.append(lv['dho'].data_ptr() + 4 * r0 * ncol)
                bbs.append(lv['bbox'].data_ptr() + 48 * r0)
                dbbs.append(lv['dbbox'].data_ptr() + 48 * r0)
                n = lo[l + 1] - lo[l]
                if n:
                    call('es_focal_loss', hos[l] + 4 * 13, ncol, kt.data_ptr() + 4 * lo[l], n, self.num_classes,
                         self.focal_gamma, self.focal_alpha, avg.data_ptr() + 4 * b, gscale, dhos[l] + 4 * 13, ncol,
                         P(partial), loss_cls.data_ptr() + 4 * b, s)
            # every ground-truth box keeps at most pts_center_threshold locations (fcaf3d_head.py:1640-1650)
            max_pos = min(lo[-1], self.pts_center_threshold * int(gts[b][0].shape[0]))
            if max_pos:
                pos_ws = torch.empty(max_pos + 1, dtype=torch.int32, device=dev)
                call('es_pos_losses', P(kt), lo[-1], P(npos), max_pos, P(pos_ws), P(pts), n_lvl, iarr(lo), parr(hos),
                     parr(bbs), parr(dhos), parr(dbbs), ncol, P(ct), P(bt), avg.data_ptr() + 4 * b, gscale, gwa,
                     loss_acc.data_ptr() + 16 * b, s)

        if on_side:
            with E.side_stream():
                for b in on_side:
                    sample_losses(b, partials[1])
        for b in on_main:
            sample_losses(b, partials[0])
        if on_side:
            E.join_side()
        s = _stream()
        # chain through exp/Scale/clamp and seed the head GEMM gradients
        for lv in levels:
            n = lv['cs'].n
            ncol = lv['ho'].d.shape[1]
            call('es_reg_decode_bwd', lv['ho'].d.data_ptr() + 4, ncol, P(lv['bbox']), P(lv['dbbox']), n,
                 P(lv['scale'].d), lv['dho'].data_ptr() + 4, ncol, P(lv['scale'].g), P(partials[0]), s)
            lv['ho'].g = lv['dho']
        eps = float(torch.finfo(torch.float32).eps)
        loss_acc = loss_acc.float()
        losses = dict(loss_center=(loss_acc[:, 0] / (avg + eps)).mean(), loss_bbox=loss_acc[:, 1].mean(),
                      loss_cls=loss_cls.mean())
        self.last_targets = [(p[2], p[3], p[4]) for p in per]
        self.last_levels = levels
        return losses
