"""GroundingHead (embodiedscan/models/dense_heads/grounding_head.py:102-824) on the MI355X kernels, for the shipped
configurations: box_coder 'baseline' or 'FCAF' with 9 regression outputs, share_pred_layer=True, sigmoid FocalLoss on the token
logits, 4-group decoupled BBoxCDLoss, HungarianAssigner3D(BinaryFocalLossCost, BBox3DL1Cost, IoU3DCost).
Per decoder layer the loss is five launches for the WHOLE batch: contrastive logits, costs + assignment (device-side
Hungarian, no D2H / scipy), labels + focal loss with its gradient, corner-Chamfer loss on the matched pairs with its
gradient -- the reference loops over samples with a host round trip each."""
import math
import os
import torch
from ... import engine as E
from ... import hip
from ...hip import P, call, farr
from ...parallel import reduce_mean
from ...registry import MODELS, TASK_UTILS
from ..layers.ground_transformer.decoder import _Lin

MATCH_BATCH = [os.environ.get('ES_MATCH_BATCH', '1') != '0']     # round 6: the decoder layers' assignments in one launch (A/B switch)


@MODELS.register_module()
class GroundingHead:
    def __init__(self, num_classes, embed_dims=256, num_pred_layer=7, num_reg_fcs=2, num_reg=9, box_coder='baseline',
                 sync_cls_avg_factor=False, decouple_bbox_loss=False, decouple_groups=3, decouple_weights=None,
                 norm_decouple_loss=False, loss_cls=None, loss_bbox=None, train_cfg=None, contrastive_cfg=None,
                 share_pred_layer=False, test_cfg=None, init_cfg=None):
        from .. import task_modules  # noqa: F401
        self.contrastive_cfg = dict(contrastive_cfg or dict(max_text_len=256))
        self.max_text_len = self.contrastive_cfg.get('max_text_len', 256)
        assert self.contrastive_cfg.get('log_scale', None) == 'auto' and self.contrastive_cfg.get('bias', False), \
            "the fused kernel implements ContrastiveEmbed(log_scale='auto', bias=True) of the shipped config"
        # the shipped configs use 9 regression outputs with either coder (configs/grounding/*.py; ..._fcaf-coder.py:64: 'FCAF')
        assert box_coder in ('baseline', 'FCAF') and num_reg == 9 and num_reg_fcs == 2, 'shipped: 9 outputs, baseline or FCAF coder'
        self.box_coder = box_coder
        assert share_pred_layer, 'shipped: share_pred_layer=True'
        loss_cls = loss_cls or {}
        assert loss_cls.get('type') == 'mmdet.FocalLoss' and loss_cls.get('use_sigmoid', False)
        self.focal_gamma, self.focal_alpha = float(loss_cls.get('gamma', 2.0)), float(loss_cls.get('alpha', 0.25))
        self.loss_cls_weight = float(loss_cls.get('loss_weight', 1.0))
        loss_bbox = loss_bbox or {}
        assert loss_bbox.get('type') == 'BBoxCDLoss' and loss_bbox.get('mode') == 'l1' and loss_bbox.get('group', 'g8') == 'g8'
        self.loss_bbox_weight = float(loss_bbox.get('loss_weight', 1.0))
        assert not norm_decouple_loss
        self.num_classes, self.embed_dims, self.num_pred_layer, self.num_reg = num_classes, embed_dims, num_pred_layer, num_reg
        self.sync_cls_avg_factor = sync_cls_avg_factor
        self.decouple_bbox_loss, self.decouple_groups = decouple_bbox_loss, decouple_groups
        self.decouple_weights = list(decouple_weights) if decouple_weights is not None else [1.0 / decouple_groups] * decouple_groups
        self.bg_cls_weight = 0
        self.train_cfg, self.test_cfg = train_cfg, test_cfg
        self.assigner = TASK_UTILS.build(train_cfg['assigner']) if train_cfg else None
        self.training = True

    def bind(self, arena, prefix='bbox_head.'):
        self.cls_bias = E.Param(arena.p[prefix + 'cls_branches.0.bias'], arena.g.get(prefix + 'cls_branches.0.bias'))
        self.reg = [_Lin(arena, prefix + f'reg_branches.0.{j}.weight', prefix + f'reg_branches.0.{j}.bias') for j in (0, 2, 4)]
        return self

    # ------------------------------------------------------------------ branches
    def reg_branch(self, x):
        """reg_branches[i] (shared): Linear-ReLU-Linear-ReLU-Linear -> Var (n, 9)"""
        h = E.relu_(self.reg[0](x))
        h = E.relu_(self.reg[1](h))
        return self.reg[2](h)

    def decode(self, points, reg):
        """_bbox_pred_to_bbox (grounding_head.py:267-363, 'baseline' :292-296 or 'FCAF' :308-363): Var (n,9) boxes from raw (n,3)
        points and the reg Var"""
        n = reg.d.shape[0]
        box = E.Var(torch.empty((n, 9), dtype=torch.float32, device=reg.d.device))
        fwd, bwd_name = (('es_ground_decode_fwd', 'es_ground_decode_bwd') if self.box_coder == 'baseline' else
                         ('es_ground_decode_fcaf_fwd', 'es_ground_decode_fcaf_bwd'))
        call(fwd, P(reg.d), reg.d.stride(0), P(points), n, P(box.d), hip.stream())

        def bwd():
            if box.g is None:
                return
            acc = 1 if reg.g is not None else 0
            if reg.g is None:
                reg.g = torch.empty_like(reg.d)
            call(bwd_name, P(reg.d), reg.d.stride(0), P(box.g), n, P(reg.g), reg.g.stride(0), acc, hip.stream())
        E.TAPE.add(bwd)
        return box

    def cls_branch(self, visual, text, B, L, T, tlen, vlen=None, want_logits=True, want_max=False):
        """ContrastiveEmbed: visual Var (B*L, E), text Var (B*T, E) -> (logits Var (B*L, T) or None, rowmax or None)"""
        dev = visual.d.device
        C = visual.d.shape[1]
        logits = E.Var(torch.empty((B * L, T), dtype=torch.float32, device=dev)) if want_logits else None
        rowmax = torch.empty(B * L, dtype=torch.float32, device=dev) if want_max else None
        call('es_contrastive_fwd', P(visual.d), B, L, P(text.d), T, C, P(tlen), P(vlen), P(self.cls_bias.d),
             P(logits.d) if logits is not None else 0, T, P(rowmax), hip.stream())
        if logits is not None:
            def bwd():
                if logits.g is None:
                    return
                s = hip.stream()
                if visual.rg:
                    acc = 1 if visual.g is not None else 0
                    if visual.g is None:
                        visual.g = torch.empty_like(visual.d)
                    gv = visual.g
                else:
                    gv, acc = None, 0
                if text.rg and text.g is None:
                    text.g = torch.zeros_like(text.d)
                ws, nws = E.ticket_ws(hip.raw('es_contrastive_bwd_workspace_floats')(B, T), visual.d)
                rec = None
                if E.DEBUG_OPS is not None:
                    rec = dict(kind='contrastive', dl=logits.g.clone(), v=visual.d.clone(), text=text.d.clone(), tlen=tlen.clone(),
                               B=B, L=L, T=T, C=C, acc=acc, dv0=gv.clone() if (gv is not None and acc) else None,
                               dtext0=text.g.clone() if text.rg else None, dbias0=self.cls_bias.g.clone())
                    E.DEBUG_OPS.append(rec)
                call('es_contrastive_bwd', P(logits.g), T, P(visual.d), B, L, P(text.d), T, C, P(tlen), P(gv), acc,
                     P(text.g) if text.rg else 0, P(self.cls_bias.g), P(ws), nws, s)
                if rec is not None:
                    rec.update(dv1=None if gv is None else gv.clone(), dtext1=text.g.clone() if text.rg else None,
                               dbias1=self.cls_bias.g.clone())
            E.TAPE.add(bwd)
        return logits, rowmax

    # ------------------------------------------------------------------ ground truth upload
    @staticmethod
    def pack_gt(batch_gt_instances_3d, T, dev):
        """-> gt_boxes (sumG,9) f32, pos_map (sumG,T) u8, gt_off (B+1) int32 (device), Gs (host list)"""
        Gs, boxes, maps = [], [], []
        for gi in batch_gt_instances_3d:
            b = getattr(gi.bboxes_3d, 'tensor', gi.bboxes_3d)
            Gs.append(int(b.shape[0]))
            boxes.append(b.detach().float().cpu().reshape(-1, 9))
            pm = gi.positive_maps.detach().cpu()
            m = torch.zeros((pm.shape[0], T), dtype=torch.uint8)
            w = min(T, pm.shape[1])
            m[:, :w] = (pm[:, :w] != 0).to(torch.uint8)
            maps.append(m)
        off = [0]
        for g in Gs:
            off.append(off[-1] + g)
        gt_boxes = torch.cat(boxes).contiguous() if boxes else torch.zeros((0, 9))
        pos_map = torch.cat(maps).contiguous() if maps else torch.zeros((0, T), dtype=torch.uint8)
        if gt_boxes.shape[0] == 0:
            gt_boxes, pos_map = torch.zeros((1, 9)), torch.zeros((1, T), dtype=torch.uint8)
        return (gt_boxes.to(dev, non_blocking=True), pos_map.to(dev, non_blocking=True),
                torch.tensor(off, dtype=torch.int32).to(dev, non_blocking=True), Gs)

    # ------------------------------------------------------------------ loss
    def loss(self, hidden_states, all_layers_pred_bboxes, text_feats, text_token_mask, batch_data_samples, tlen=None):
        """grounding_head.py:606-822.  hidden_states / all_layers_pred_bboxes: lists over decoder layers of Vars
        (B*Q, E) / (B*Q, 9); text_feats Var (B*T, E); text_token_mask (B, T) bool (prefix masks)."""
        gis = [ds.gt_instances_3d for ds in batch_data_samples]
        B = len(gis)
        T = text_token_mask.shape[1]
        dev = hidden_states[0].d.device
        Q = hidden_states[0].d.shape[0] // B
        if tlen is None:
            tlen = text_token_mask.sum(1).to(torch.int32).to(dev)
        gt_boxes, pos_map, gt_off, Gs = self.pack_gt(gis, T, dev)
        gt_off_host = [0]
        for g_ in Gs:
            gt_off_host.append(gt_off_host[-1] + g_)
        Gmax = max(Gs) if Gs else 0
        assert Gmax <= Q, 'more target boxes than queries'
        n_pos = sum(Gs)
        # cls_avg_factor = max(reduce_mean(num_total_pos + num_total_neg * bg_cls_weight(=0)), 1): identical for all layers
        avg = torch.tensor([float(n_pos)], dtype=torch.float32, device=dev)
        if self.sync_cls_avg_factor:
            avg = reduce_mean(avg)
        avg = avg.clamp(min=1.0).contiguous()
        gw = [w * self.loss_bbox_weight for w in self.decouple_weights]
        if not self.decouple_bbox_loss:
            gw = [0., 0., 0., self.loss_bbox_weight]
        elif self.decouple_groups == 3:
            gw = gw[:3] + [0.]
        gwa = farr(gw)
        eps = float(torch.finfo(torch.float32).eps)
        s = hip.stream()
        losses, self.last = {}, []
        L = len(hidden_states)
        logits_l = [self.cls_branch(hidden_states[l], text_feats, B, Q, T, tlen)[0] for l in range(L)]
        q2g_all = None
        if MATCH_BATCH[0] and L > 1:
            # round 6: the assignments of the L decoder layers are independent -- ONE cost + assignment launch over L * B "samples" (the targets
            # tiled L times) instead of L dependent pairs on the main stream (0.2 ms each: a few hundred f64 box-IoU threads, far from filling
            # the chip); same kernels, same per-(layer, sample) arithmetic
            sumG = int(gt_off_host[-1])
            all_logits = torch.cat([lg.d for lg in logits_l]).view(L * B, Q, T)
            all_boxes = torch.cat([bx.d for bx in all_layers_pred_bboxes]).view(L * B, Q, 9)
            off_L = torch.tensor([l * sumG + o for l in range(L) for o in gt_off_host[:-1]] + [L * sumG], dtype=torch.int32).to(dev, non_blocking=True)
            q2g_all = self.assigner.match(all_logits, all_boxes, gt_boxes.repeat(L, 1), pos_map.repeat(L, 1), off_L, Gmax, tlen.repeat(L), s)
            self.assigner.last_cost = self.assigner.last_cost.view(L, B, max(Gmax, 1), Q)[-1]
        # one zero-filled accumulator pair / gradient buffer for all layers, the loss scalars derived once behind the loop (round 6: the loop
        # used to issue ~ 8 one-element torch launches per layer)
        lsum_all = torch.zeros(L, dtype=torch.float64, device=dev)
        lbox_all = torch.zeros(L, dtype=torch.float64, device=dev)         # (f64 accumulators: order-independent, see csrc/losses.hip)
        bgrad_all = torch.zeros((L,) + tuple(all_layers_pred_bboxes[0].d.shape), dtype=torch.float32, device=dev)
        for l in range(L):
            logits = logits_l[l]
            boxes = all_layers_pred_bboxes[l]
            if q2g_all is not None:
                q2g = q2g_all[l * B:(l + 1) * B]
            else:
                q2g = self.assigner.match(logits.d.view(B, Q, T), boxes.d.view(B, Q, 9), gt_boxes, pos_map, gt_off, Gmax, tlen, s)
            if getattr(self, 'force_assign', None) is not None:      # test hook (teacher forcing): the oracle's assignment of layer l
                free, q2g = q2g, self.force_assign[l].to(device=dev, dtype=torch.int32).reshape(B, Q).contiguous()
            logits.g = torch.empty_like(logits.d)
            call('es_ground_focal', P(logits.d), T, B, Q, P(q2g), P(pos_map), P(gt_off), P(tlen), T, self.focal_alpha,
                 self.focal_gamma, P(avg), self.loss_cls_weight, P(logits.g), lsum_all.data_ptr() + 8 * l, s)
            boxes.g = bgrad_all[l]
            if n_pos:
                call('es_box_cd_pairs', P(boxes.d), P(q2g), B, Q, P(gt_boxes), P(gt_off), n_pos, 1.0, gwa, P(boxes.g), lbox_all.data_ptr() + 8 * l, s)
            self.last.append(dict(logits=logits, boxes=boxes, q2g=q2g))
            if getattr(self, 'force_assign', None) is not None:
                self.last[-1]['q2g_free'] = free
        loss_cls_all = lsum_all.float() / (avg + eps) * self.loss_cls_weight
        loss_box_all = lbox_all.float()
        for l in range(L):
            name = '' if l == L - 1 else f'd{l}.'
            losses[name + 'loss_cls'] = loss_cls_all[l]
            losses[name + 'loss_bbox'] = loss_box_all[l]
        out = dict(loss_cls=losses['loss_cls'], loss_bbox=losses['loss_bbox'])
        out.update({k: v for k, v in losses.items() if k.startswith('d')})
        return out

    # ------------------------------------------------------------------ predict
    def predict(self, hidden_states, all_layers_pred_bboxes, text_feats, text_token_mask, batch_data_samples, tlen=None):
        """grounding_head.py:455-604: per sample InstanceData(bboxes_3d, scores_3d, target_scores_3d) from the LAST layer"""
        from ...structures import EulerDepthInstance3DBoxes, InstanceData
        B = len(batch_data_samples)
        T = text_token_mask.shape[1]
        dev = hidden_states[-1].d.device
        Q = hidden_states[-1].d.shape[0] // B
        if tlen is None:
            tlen = text_token_mask.sum(1).to(torch.int32).to(dev)
        _, rowmax = self.cls_branch(hidden_states[-1], text_feats, B, Q, T, tlen, want_logits=False, want_max=True)
        scores = torch.sigmoid(rowmax).view(B, Q)               # max over tokens of sigmoid == sigmoid of the max
        boxes = all_layers_pred_bboxes[-1].d.view(B, Q, 9)
        out = []
        for b in range(B):
            out.append(InstanceData(bboxes_3d=EulerDepthInstance3DBoxes(boxes[b]), scores_3d=scores[b], target_scores_3d=scores[b]))
        return out
