"""Generate tests/golden/ground_*.npz by running the REFERENCE's own grounding code.

Run in the build container only (needs /root/reference):   python -m oracle.make_golden_ground
TEST INFRASTRUCTURE (mechanism: oracle/make_golden.py).  Reference entry points exercised (file:line):
  ContrastiveEmbed.forward                          models/dense_heads/grounding_head.py:62-99
  GroundingHead._bbox_pred_to_bbox ('baseline', 9)  models/dense_heads/grounding_head.py:267-296
  GroundingHead._bbox_pred_to_bbox ('FCAF', 9)      models/dense_heads/grounding_head.py:308-363  (-> ground_coder_fcaf.npz)
  GroundingHead.loss_by_feat_single (+ get_targets / _get_targets_single)   :226-265,365-425,686-822
  HungarianAssigner3D.assign                        models/task_modules/assigners/hungarian_assigner.py:56-138
  BinaryFocalLossCost / BBox3DL1Cost / IoU3DCost    models/losses/match_cost.py:49-75,95-113,213-265
  PositionEmbeddingLearned.forward                  models/layers/ground_transformer/decoder.py:20-34
  BBoxCDLoss                                        models/losses/chamfer_distance.py:265-285
Un-vendored pieces bound to restatements: pytorch3d.ops.box3d_overlap -> oracle.grounding (exact polyhedral IoU, itself
checked against scipy's qhull in tests), mmdet.FocalLoss -> py_sigmoid_focal_loss restated, scipy is real."""
import os
import types
import numpy as np
import torch


def main(out_dir=None):
    from . import _ref_stubs
    _ref_stubs.install()
    from . import grounding as OG
    import embodiedscan.structures.bbox_3d.euler_box3d as EB

    def box3d_overlap(c1, c2, eps=1e-4):            # the reference passes CORNERS; recover nothing: compute from them
        raise RuntimeError('unused')
    from embodiedscan.structures import EulerDepthInstance3DBoxes

    def overlaps(cls, boxes1, boxes2, mode='iou', eps=1e-4):
        return OG.overlaps(boxes1.tensor, boxes2.tensor)
    EB.EulerInstance3DBoxes.overlaps = classmethod(overlaps)      # box3d_overlap (pytorch3d) -> the oracle's exact IoU
    from embodiedscan.models.dense_heads.grounding_head import ContrastiveEmbed, GroundingHead
    from embodiedscan.models.task_modules.assigners.hungarian_assigner import HungarianAssigner3D
    from embodiedscan.models.losses.match_cost import BBox3DL1Cost, BinaryFocalLossCost, IoU3DCost
    from embodiedscan.models.losses.chamfer_distance import BBoxCDLoss
    from embodiedscan.models.layers.ground_transformer.decoder import PositionEmbeddingLearned
    from mmengine.structures import InstanceData
    out_dir = out_dir or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')
    g = torch.Generator().manual_seed(20240926)
    rnd = lambda *s, lo=-1., hi=1.: torch.rand(*s, generator=g) * (hi - lo) + lo

    B, Q, T, E = 2, 12, 9, 16
    tlens = [9, 6]
    mask = torch.stack([torch.arange(T) < t for t in tlens])
    hidden, text = torch.randn(B, Q, E, generator=g), torch.randn(B, T, E, generator=g)
    ce = ContrastiveEmbed(max_text_len=32, log_scale='auto', bias=True)
    cls = ce(hidden, text, mask)
    rec = dict(hidden=hidden.numpy(), text=text.numpy(), mask=mask.numpy(), cls=cls.detach().numpy(), ce_bias=ce.bias.detach().numpy())

    # box coder
    pts, pred = rnd(B, Q, 3, lo=-2, hi=2), torch.cat([rnd(B, Q, 3, lo=-.5, hi=.5), rnd(B, Q, 3, lo=-5, hi=1), rnd(B, Q, 3, lo=-3, hi=3)], -1)
    head = types.SimpleNamespace(box_coder='baseline')
    boxes = GroundingHead._bbox_pred_to_bbox(head, pts, pred)
    rec.update(points=pts.numpy(), reg=pred.numpy(), boxes=boxes.numpy())

    # ground truth: overlapping targets so that the IoU cost matters
    Gs = [3, 1]
    gtb, pms = [], []
    for b in range(B):
        pick = torch.randperm(Q, generator=g)[:Gs[b]]
        gb = boxes[b, pick].clone()
        gb[:, :3] += rnd(Gs[b], 3, lo=-.15, hi=.15)
        gb[:, 3:6] *= rnd(Gs[b], 3, lo=.8, hi=1.25)
        gb[:, 6:] += rnd(Gs[b], 3, lo=-.2, hi=.2)
        gtb.append(gb)
        pm = torch.zeros(Gs[b], 32)
        for k in range(Gs[b]):
            a = int(torch.randint(1, tlens[b] - 2, (1,), generator=g))
            pm[k, a:a + 2] = 1
        pms.append(pm)
    gis = [InstanceData(bboxes_3d=EulerDepthInstance3DBoxes(gtb[b]), labels_3d=torch.zeros(Gs[b], dtype=torch.long),
                        positive_maps=pms[b], text_token_mask=mask[b][None].repeat(Gs[b], 1)) for b in range(B)]
    assigner = HungarianAssigner3D([dict(type='x')]) if False else HungarianAssigner3D.__new__(HungarianAssigner3D)
    assigner.match_costs = [BinaryFocalLossCost(weight=1.0), BBox3DL1Cost(weight=2.0), IoU3DCost(weight=2.0)]
    costs, ginds = [], []
    for b in range(B):
        pi = InstanceData(scores_3d=cls[b].detach(), bboxes_3d=EulerDepthInstance3DBoxes(boxes[b]))
        costs.append(torch.stack([m(pred_instances=pi, gt_instances=gis[b]) if not isinstance(m, IoU3DCost) else m(pi, gis[b])
                                  for m in assigner.match_costs]).numpy())
        ginds.append(assigner.assign(pi, gis[b]).gt_inds.numpy())
    for b in range(B):
        rec[f'gt_boxes{b}'], rec[f'pos_map{b}'], rec[f'costs{b}'], rec[f'gt_inds{b}'] = gtb[b].numpy(), pms[b].numpy(), costs[b], ginds[b]

    # loss_by_feat_single through a stand-in self (mmdet FocalLoss restated; everything else is the reference's code)
    def focal(pred_, target, weight=None, avg_factor=None):
        loss = OG.py_sigmoid_focal_loss_sum(pred_, target)           # weight == 1 everywhere
        return loss / (avg_factor + float(torch.finfo(torch.float32).eps))
    hs = types.SimpleNamespace(assigner=assigner, max_text_len=32, text_masks=mask, bg_cls_weight=0, sync_cls_avg_factor=True,
                               loss_cls=focal, loss_bbox=BBoxCDLoss(mode='l1', loss_weight=1.0, group='g8'),
                               decouple_bbox_loss=True, decouple_groups=4, decouple_weights=[0.2, 0.2, 0.2, 0.4],
                               norm_decouple_loss=False)
    hs._get_targets_single = types.MethodType(GroundingHead._get_targets_single, hs)
    hs.get_targets = types.MethodType(GroundingHead.get_targets, hs)
    c = cls.detach().clone().requires_grad_(True)
    bx = boxes.detach().clone().requires_grad_(True)
    lc, lb = GroundingHead.loss_by_feat_single(hs, c, bx, gis, [{}] * B)
    (lc + lb).backward()
    rec.update(loss_cls=lc.item(), loss_bbox=lb.item(), dcls=torch.nan_to_num(c.grad, 0, 0, 0).numpy(), dboxes=bx.grad.numpy())

    # learned position embedding (train-mode BatchNorm1d over all B*N positions)
    torch.manual_seed(3)
    pe = PositionEmbeddingLearned(9, 16)
    pe.train()
    x = rnd(2, 7, 9)
    y = pe(x)
    rec.update({'pe.' + k: v.detach().numpy() for k, v in pe.state_dict().items() if 'num_batches' not in k})
    rec.update(pe_x=x.numpy(), pe_y=y.detach().numpy())
    np.savez_compressed(os.path.join(out_dir, 'ground_head.npz'), **rec)
    print('wrote ground_head.npz to', out_dir)

    # box_coder='FCAF', 9 outputs (grounding_head.py:308-363; configs/grounding/..._fcaf-coder.py:64): forward and the gradient
    # w.r.t. the raw regression output (the reference writes exp().clamp() IN PLACE into bbox_pred -- autograd sees through it)
    g2 = torch.Generator().manual_seed(20250925)
    r2 = lambda *s, lo=-1., hi=1.: torch.rand(*s, generator=g2) * (hi - lo) + lo
    pts2 = r2(B, Q, 3, lo=-2, hi=2)
    pred2 = torch.cat([r2(B, Q, 6, lo=-5, hi=1), r2(B, Q, 3, lo=-3, hi=3)], -1).requires_grad_(True)
    head2 = types.SimpleNamespace(box_coder='FCAF')
    boxes2 = GroundingHead._bbox_pred_to_bbox(head2, pts2, pred2 * 1.0)
    gb2 = torch.randn(B, Q, 9, generator=g2)
    (boxes2 * gb2).sum().backward()
    np.savez_compressed(os.path.join(out_dir, 'ground_coder_fcaf.npz'), points=pts2.numpy(), reg=pred2.detach().numpy(),
                        boxes=boxes2.detach().numpy(), dboxes=gb2.numpy(), dreg=pred2.grad.numpy())
    print('wrote ground_coder_fcaf.npz to', out_dir)


if __name__ == '__main__':
    main()
