"""HungarianAssigner3D and its match costs (embodiedscan/models/task_modules/assigners/hungarian_assigner.py:14-138,
embodiedscan/models/losses/match_cost.py:49-265) as registry entries.  On the MI355X path the three costs of the shipped
grounding config (BinaryFocalLossCost, BBox3DL1Cost, IoU3DCost) and the assignment itself run inside ONE device launch
per decoder layer (csrc/ground.hip: es_ground_match), so these classes only carry the configured weights; `assign()` is
the single-sample entry with the reference's return convention (gt_inds: 0 = background, k > 0 = matched to box k-1)."""
import torch
from ...hip import P, call
from ...registry import TASK_UTILS


@TASK_UTILS.register_module()
class BinaryFocalLossCost:
    def __init__(self, alpha=0.25, gamma=2, eps=1e-12, binary_input=False, weight=1.):
        assert alpha == 0.25 and gamma == 2 and eps == 1e-12, 'the fused kernel implements the shipped constants'
        self.weight = weight


@TASK_UTILS.register_module()
class BBox3DL1Cost:
    def __init__(self, weight=1.):
        self.weight = weight


@TASK_UTILS.register_module()
class IoU3DCost:
    def __init__(self, weight):
        self.weight = weight


@TASK_UTILS.register_module()
class HungarianAssigner3D:
    def __init__(self, match_costs):
        if isinstance(match_costs, dict):
            match_costs = [match_costs]
        assert len(match_costs) > 0, 'match_costs must not be a empty list.'
        self.match_costs = [TASK_UTILS.build(m) for m in match_costs]
        w = {type(m).__name__: m.weight for m in self.match_costs}
        unknown = set(w) - {'BinaryFocalLossCost', 'BBox3DL1Cost', 'IoU3DCost'}
        assert not unknown, f'match costs without a device implementation: {unknown}'
        self.w_cls, self.w_l1, self.w_iou = w.get('BinaryFocalLossCost', 0.), w.get('BBox3DL1Cost', 0.), w.get('IoU3DCost', 0.)

    def match(self, logits, boxes, gt_boxes, pos_map, gt_off, Gmax, tlen, stream):
        """batched device assignment.  logits (B,Q,T) f32, boxes (B,Q,9), gt_boxes (sumG,9), pos_map (sumG,T) u8,
        gt_off (B+1) / tlen (B) int32 on the device -> q2g (B,Q) int32: matched box (local index) or -1."""
        B, Q, T = logits.shape
        dev = logits.device
        q2g = torch.empty((B, Q), dtype=torch.int32, device=dev)
        cost = torch.empty(B * max(Gmax, 1) * Q, dtype=torch.float64, device=dev)
        work = torch.empty(B * (Gmax + 2 * Q), dtype=torch.float64, device=dev)
        iwork = torch.empty(B * (4 * Q + 2 * Gmax), dtype=torch.int32, device=dev)
        call('es_ground_match', P(logits), T, P(boxes), B, Q, P(gt_boxes), P(pos_map), P(gt_off), int(Gmax), P(tlen), T,
             float(self.w_cls), float(self.w_l1), float(self.w_iou), P(cost), P(work), P(iwork), P(q2g), stream)
        self.last_cost = cost.view(B, max(Gmax, 1), Q)
        return q2g

    def assign(self, pred_instances_3d, gt_instances_3d, eps=1e-7):
        """single-sample protocol of the reference: returns gt_inds (num_preds,) long"""
        scores, boxes = pred_instances_3d.scores_3d, pred_instances_3d.bboxes_3d
        boxes = getattr(boxes, 'tensor', boxes)
        gtb = getattr(gt_instances_3d.bboxes_3d, 'tensor', gt_instances_3d.bboxes_3d)
        dev = scores.device
        Q, G = scores.shape[0], gtb.shape[0]
        if G == 0 or Q == 0:
            return torch.zeros(Q, dtype=torch.long, device=dev) if G == 0 else torch.full((Q,), -1, dtype=torch.long, device=dev)
        tmask = gt_instances_3d.text_token_mask[0].to(dev).bool()
        tl = int(tmask.sum())
        assert bool(tmask[:tl].all()), 'text token masks are prefix masks'
        T = scores.shape[1]
        pm = gt_instances_3d.positive_maps.to(dev)[:, :T].bool().to(torch.uint8).contiguous()
        q2g = self.match(scores.float().contiguous()[None], boxes.float().contiguous()[None], gtb.float().contiguous().to(dev), pm,
                         torch.tensor([0, G], dtype=torch.int32, device=dev), G, torch.tensor([tl], dtype=torch.int32, device=dev),
                         torch.cuda.current_stream(dev).cuda_stream)[0]
        return (q2g + 1).long()
