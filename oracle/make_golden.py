"""Generate tests/golden/*.npz by running the REFERENCE's own pure-PyTorch functions.

Run in the build container only (needs /root/reference):   python -m oracle.make_golden
The reference is imported from where it lies, with oracle/_ref_stubs.py standing in for
the absent third-party packages.  The outputs are small committed fixtures; the GPU box
and the tests never read /root/reference.  TEST INFRASTRUCTURE.

Reference entry points exercised (file:line):
  rotation_3d_in_euler                    structures/bbox_3d/utils.py:32-86
  EulerDepthInstance3DBoxes.corners       structures/bbox_3d/euler_box3d.py:142-184
  points_img2cam / ConvertRGBDToPoints    structures/bbox_3d/utils.py:335-368, datasets/transforms/points.py:30-81
  AggregateMultiViewPoints.transform      datasets/transforms/multiview.py:139-169
  Det3DDataPreprocessor.preprocess_img    models/data_preprocessors/data_preprocessor.py:249-264
  batch_point_sample                      models/layers/fusion_layers/point_fusion.py:208-311
  FCAF3DHeadRotMat.get_targets            models/dense_heads/fcaf3d_head.py:1578-1664
  FCAF3DHeadRotMat._bbox_pred_to_bbox     models/dense_heads/fcaf3d_head.py:1454-1525
  BBoxCDLoss.forward                      models/losses/chamfer_distance.py:265-285
"""
import os
import types
import numpy as np
import torch


def main(out_dir=None):
    from . import _ref_stubs
    _ref_stubs.install()
    from embodiedscan.structures import EulerDepthInstance3DBoxes, rotation_3d_in_euler
    from embodiedscan.structures.bbox_3d.utils import points_img2cam
    from embodiedscan.models.layers.fusion_layers.point_fusion import batch_point_sample
    from embodiedscan.models.dense_heads.fcaf3d_head import FCAF3DHeadRotMat
    from embodiedscan.models.losses.chamfer_distance import BBoxCDLoss
    from embodiedscan.models.data_preprocessors.data_preprocessor import Det3DDataPreprocessor
    from embodiedscan.datasets.transforms.points import ConvertRGBDToPoints
    from embodiedscan.datasets.transforms.multiview import AggregateMultiViewPoints

    out_dir = out_dir or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')
    os.makedirs(out_dir, exist_ok=True)
    g = torch.Generator().manual_seed(20240924)

    def rnd(*s, lo=-1., hi=1.):
        return torch.rand(*s, generator=g) * (hi - lo) + lo

    # ---- euler rotation + corners
    pts, ang = rnd(5, 7, 3, lo=-2, hi=2), rnd(5, 3, lo=-3.1, hi=3.1)
    boxes = torch.cat([rnd(6, 3, lo=-3, hi=3), rnd(6, 3, lo=.2, hi=2.), rnd(6, 3, lo=-3.1, hi=3.1)], 1)
    np.savez_compressed(os.path.join(out_dir, 'euler.npz'), points=pts.numpy(), angles=ang.numpy(),
             rotated=rotation_3d_in_euler(pts, ang).numpy(), boxes=boxes.numpy(),
             corners=EulerDepthInstance3DBoxes(boxes).corners.numpy())

    # ---- A1 / A3: depth -> camera points -> global points
    H, W = 12, 16
    depth = rnd(H, W, lo=0.3, hi=5.0)
    depth[rnd(H, W) > 0.6] = 0
    K = torch.tensor([[14.5, 0, 7.5, 0], [0, 14.5, 5.5, 0], [0, 0, 1, 0], [0, 0, 0, 1.]])
    res = ConvertRGBDToPoints(coord_type='CAMERA').transform(dict(depth_img=depth.numpy(), depth_cam2img=K.numpy()))
    cam_pts = res['points'].tensor.clone()
    a = rnd(3, lo=-3, hi=3)
    from pytorch3d.transforms import euler_angles_to_matrix
    g2c = torch.eye(4)
    g2c[:3, :3] = euler_angles_to_matrix(a, 'ZXY')
    g2c[:3, 3] = rnd(3, lo=-2, hi=2)
    agg = AggregateMultiViewPoints(coord_type='DEPTH').transform(
        dict(points=[res['points']], depth2img=dict(extrinsic=[g2c.numpy()])))
    np.savez_compressed(os.path.join(out_dir, 'unproject.npz'), depth=depth.numpy(), cam2img=K.numpy(), cam_points=cam_pts.numpy(),
             global2cam=g2c.numpy(), global_points=agg['points'].tensor.numpy())

    # ---- A18 image normalisation
    img = torch.randint(0, 256, (3, 10, 14), generator=g, dtype=torch.uint8)
    fake = types.SimpleNamespace(_channel_conversion=True, _enable_normalize=True,
                                 mean=torch.tensor([123.675, 116.28, 103.53]).view(-1, 1, 1),
                                 std=torch.tensor([58.395, 57.12, 57.375]).view(-1, 1, 1))
    np.savez_compressed(os.path.join(out_dir, 'preprocess_img.npz'), img=img.numpy(),
             out=Det3DDataPreprocessor.preprocess_img(fake, img).numpy())

    # ---- A8 batch_point_sample (two cases: plain, augmented + flip + crop)
    for name, aug in (('point_sample_plain', False), ('point_sample_aug', True)):
        V, C, h, w = 3, 5, 12, 16
        pad_shape, img_shape = (96, 128), (90, 120)
        feats = rnd(V, C, h, w)
        n = 400
        points = torch.cat([rnd(n, 2, lo=-3, hi=3), rnd(n, 1, lo=0, hi=2.5)], 1)
        proj = []
        for v in range(V):
            e = torch.eye(4)
            e[:3, :3] = euler_angles_to_matrix(torch.tensor([rnd(1).item() * 3, -1.4 + rnd(1).item() * .3, rnd(1).item() * .2]), 'ZXY')
            e[:3, 3] = rnd(3, lo=-1, hi=1)
            Kv = torch.tensor([[100., 0, 79.5, 0], [0, 100., 59.5, 0], [0, 0, 1, 0], [0, 0, 0, 1]])
            proj.append(Kv @ e)
        proj = torch.stack(proj)
        meta = dict()
        sf, off, flip = torch.tensor([0.75, 0.75]), 0, False
        if aug:
            rot = euler_angles_to_matrix(torch.tensor([0.07, 0., 0.]), 'ZXY')
            meta = dict(pcd_rotation=rot.t().numpy(), pcd_scale_factor=1.05, pcd_trans=np.array([.1, -.05, .02], np.float32),
                        pcd_horizontal_flip=True, pcd_vertical_flip=False, transformation_3d_flow=['HF', 'R', 'S', 'T'])
            off, flip = torch.tensor([3., 2.]), True
        out = batch_point_sample(meta, feats, points, proj, 'DEPTH', sf, off, flip, pad_shape, img_shape, aligned=False)
        np.savez_compressed(os.path.join(out_dir, name + '.npz'), feats=feats.numpy(), points=points.numpy(), proj=proj.numpy(),
                 scale_factor=sf.numpy(), crop_offset=np.asarray(off, np.float32) * np.ones(2, np.float32), flip=flip,
                 pad_shape=np.array(pad_shape), img_shape=np.array(img_shape), out=out.numpy(),
                 pcd_rotation=np.asarray(meta.get('pcd_rotation', np.eye(3)), np.float32),
                 pcd_scale_factor=np.float32(meta.get('pcd_scale_factor', 1.)),
                 pcd_trans=np.asarray(meta.get('pcd_trans', np.zeros(3)), np.float32),
                 hflip=meta.get('pcd_horizontal_flip', False), vflip=meta.get('pcd_vertical_flip', False),
                 flow=np.array(meta.get('transformation_3d_flow', []), dtype='U2'))

    # ---- A12 get_targets
    fake = types.SimpleNamespace(pts_assign_threshold=27, pts_center_threshold=18,
                                 _get_face_distances=FCAF3DHeadRotMat._get_face_distances,
                                 _get_centerness=FCAF3DHeadRotMat._get_centerness)
    for name, nb in (('get_targets', 7), ('get_targets_empty', 0)):
        lv = []
        for ts in (8, 16, 32, 64):
            r = torch.arange(-160, 160, ts)
            gx, gy, gz = torch.meshgrid(r, r, torch.arange(0, 192, ts), indexing='ij')
            p = torch.stack([gx.reshape(-1), gy.reshape(-1), gz.reshape(-1)], 1).float() * 0.01
            keep = torch.rand(len(p), generator=g) < 0.7
            lv.append(p[keep])
        gtb = torch.cat([rnd(nb, 2, lo=-1.3, hi=1.3), rnd(nb, 1, lo=.5, hi=1.3), rnd(nb, 3, lo=.3, hi=1.8),
                         rnd(nb, 1, lo=-3.1, hi=3.1), rnd(nb, 2, lo=-.2, hi=.2)], 1)
        gtl = torch.randint(0, 284, (nb,), generator=g)
        ct, bt, kt = FCAF3DHeadRotMat.get_targets(fake, [p.clone() for p in lv], EulerDepthInstance3DBoxes(gtb), gtl)
        np.savez_compressed(os.path.join(out_dir, name + '.npz'), **{f'points{i}': p.numpy() for i, p in enumerate(lv)},
                 gt_boxes=gtb.numpy(), gt_labels=gtl.numpy(), center_targets=ct.numpy(), bbox_targets=bt.numpy(),
                 cls_targets=kt.numpy())

    # ---- A13 box coder + A15 corner chamfer loss
    n = 64
    pred = torch.cat([rnd(n, 6, lo=.05, hi=1.5), rnd(n, 6, lo=-1, hi=1)], 1)
    pp = rnd(n, 3, lo=-3, hi=3)
    dec = FCAF3DHeadRotMat._bbox_pred_to_bbox(pp, pred)
    tgt = torch.cat([rnd(n, 3, lo=-3, hi=3), rnd(n, 3, lo=.2, hi=2.), rnd(n, 3, lo=-3.1, hi=3.1)], 1)
    crit = BBoxCDLoss(mode='l1', group='g8', loss_weight=1.0)
    np.savez_compressed(os.path.join(out_dir, 'box_coder_cdloss.npz'), points=pp.numpy(), pred=pred.numpy(), decoded=dec.numpy(),
             target=tgt.numpy(), loss=crit(dec, tgt).numpy(),
             loss_none=crit(dec, tgt, reduction_override='none').numpy())
    # ---- A14-A16 composition of the per-sample loss: the reference's own _loss_by_feat_single (targets, positive
    # selection, avg_factor, decoupled corner loss, empty-positive branch) driven through a stand-in `self`.  Only the
    # two mmdet criteria are restated (mmdet is absent): FocalLoss(use_sigmoid, gamma 2, alpha .25) and
    # CrossEntropyLoss(use_sigmoid) with mmdet's weight_reduce_loss rule  sum / (avg_factor + float32 eps).
    import torch.nn.functional as Fn
    eps = torch.finfo(torch.float32).eps

    def focal(pred, target, avg_factor):
        t = Fn.one_hot(target.clamp(min=0), pred.shape[1] + 1)[:, :pred.shape[1]].float()
        t = t * (target >= 0)[:, None]                      # label -1 (background): all-negative row
        p = pred.sigmoid()
        pt = (1 - p) * t + p * (1 - t)
        w = (0.25 * t + 0.75 * (1 - t)) * pt.pow(2.0)
        return (Fn.binary_cross_entropy_with_logits(pred, t, reduction='none') * w).sum() / (avg_factor + eps)

    def bce(pred, target, avg_factor):
        return Fn.binary_cross_entropy_with_logits(pred, target.float(), reduction='none').mean(1).sum() / (avg_factor + eps)

    for name, nb in (('loss_single', 6), ('loss_single_empty', 0)):
        lv = []
        for ts in (8, 16, 32, 64):
            r = torch.arange(-96, 96, ts)
            gx, gy, gz = torch.meshgrid(r, r, torch.arange(0, 96, ts), indexing='ij')
            p = torch.stack([gx.reshape(-1), gy.reshape(-1), gz.reshape(-1)], 1).float() * 0.01
            lv.append(p[torch.rand(len(p), generator=g) < 0.6])
        gtb = torch.cat([rnd(nb, 2, lo=-.7, hi=.7), rnd(nb, 1, lo=.3, hi=.7), rnd(nb, 3, lo=.3, hi=1.2),
                         rnd(nb, 1, lo=-3.1, hi=3.1), rnd(nb, 2, lo=-.2, hi=.2)], 1)
        gtl = torch.randint(0, 24, (nb,), generator=g)        # 24 classes keep the fixture small; the code is class-count agnostic
        cen = [rnd(len(p), 1) for p in lv]
        box = [torch.cat([rnd(len(p), 6, lo=.05, hi=1.2), rnd(len(p), 6, lo=-1, hi=1)], 1) for p in lv]
        cls = [rnd(len(p), 24, lo=-6., hi=0.5).half().float() for p in lv]     # stored as f16: round first
        me = types.SimpleNamespace(pts_assign_threshold=27, pts_center_threshold=18,
                                   _get_face_distances=FCAF3DHeadRotMat._get_face_distances,
                                   _get_centerness=FCAF3DHeadRotMat._get_centerness,
                                   _bbox_pred_to_bbox=FCAF3DHeadRotMat._bbox_pred_to_bbox,
                                   cls_loss=focal, center_loss=bce, bbox_loss=BBoxCDLoss(mode='l1', group='g8', loss_weight=1.0),
                                   decouple_bbox_loss=True, decouple_groups=4, decouple_weights=[0.2, 0.2, 0.2, 0.4],
                                   norm_decouple_loss=False)
        me.get_targets = lambda pts, b, l, me=me: FCAF3DHeadRotMat.get_targets(me, pts, b, l)
        lc, lb, lk = FCAF3DHeadRotMat._loss_by_feat_single(me, [c.clone() for c in cen], [b.clone() for b in box],
                                                          [c.clone() for c in cls], [p.clone() for p in lv],
                                                          EulerDepthInstance3DBoxes(gtb), gtl, {})
        np.savez_compressed(os.path.join(out_dir, name + '.npz'), gt_boxes=gtb.numpy(), gt_labels=gtl.numpy(),
                            losses=np.array([float(lc), float(lb), float(lk)], np.float64),
                            **{f'points{i}': p.numpy() for i, p in enumerate(lv)},
                            **{f'center{i}': c.numpy() for i, c in enumerate(cen)},
                            **{f'bbox{i}': b.numpy() for i, b in enumerate(box)},
                            **{f'cls{i}': c.numpy().astype(np.float16) for i, c in enumerate(cls)})
    # ---- N1 predict wrapper: the reference's _predict_by_feat_single + _single_scene_multiclass_nms (per-level top-k,
    # score threshold, per-class loop, output layout).  mmcv.ops.nms3d is CUDA-only and absent: the module-level name is
    # bound to the oracle's restated greedy rotated-BEV NMS, so the suppression rule itself stays unpinned; everything
    # around it is the reference's code -- including its quirk that only (x,y,z,dx,dy,dz,alpha) survive NMS and the Euler
    # box constructor pads beta = gamma = 0.
    import embodiedscan.models.dense_heads.fcaf3d_head as ref_head
    from oracle.predict import nms3d as oracle_nms3d
    ref_head.nms3d = lambda b, sc, thr: oracle_nms3d(b.detach(), sc.detach(), thr)
    me = types.SimpleNamespace(test_cfg=types.SimpleNamespace(nms_pre=300, iou_thr=0.5, score_thr=0.05),
                               _bbox_pred_to_bbox=FCAF3DHeadRotMat._bbox_pred_to_bbox)
    me._single_scene_multiclass_nms = lambda b, sc, m, me=me: FCAF3DHeadRotMat._single_scene_multiclass_nms(me, b, sc, m)
    lv = []
    for n_l in (900, 500, 200, 40):
        pts = rnd(n_l, 3, lo=-2.5, hi=2.5)
        cen = rnd(n_l, 1, lo=-1., hi=3.)
        box = torch.cat([rnd(n_l, 6, lo=.1, hi=.9), rnd(n_l, 6, lo=-1, hi=1)], 1)
        cls = rnd(n_l, 12, lo=-7., hi=-1.)
        hot = torch.rand(n_l, generator=g) < 0.15
        cls[hot, torch.randint(0, 12, (int(hot.sum()),), generator=g)] += 6.0
        lv.append((cen, box, cls, pts))
    res = FCAF3DHeadRotMat._predict_by_feat_single(me, [l[0] for l in lv], [l[1] for l in lv], [l[2] for l in lv],
                                                   [l[3] for l in lv], dict(box_type_3d=EulerDepthInstance3DBoxes))
    np.savez_compressed(os.path.join(out_dir, 'predict_single.npz'), nms_pre=300, iou_thr=0.5, score_thr=0.05,
                        boxes=res.bboxes_3d.tensor.numpy(), scores=res.scores_3d.numpy(), labels=res.labels_3d.numpy(),
                        **{f'center{i}': l[0].numpy() for i, l in enumerate(lv)}, **{f'bbox{i}': l[1].numpy() for i, l in enumerate(lv)},
                        **{f'cls{i}': l[2].numpy() for i, l in enumerate(lv)}, **{f'points{i}': l[3].numpy() for i, l in enumerate(lv)})
    # ---- A3 augmentation of points AND boxes: RandomFlip3D.random_flip_data_3d / GlobalRotScaleTrans._rot_bbox_points,
    # _scale_bbox_points, _trans_bbox_points (augmentation.py:140-168,322-420) are thin wrappers over the box / point
    # classes' flip / rotate / scale / translate, which are called here directly with fixed "random" decisions.
    for name, hf, vf in (('augment_hv', True, True), ('augment_h', True, False), ('augment_none', False, False)):
        pts = rnd(200, 3, lo=-3, hi=3)
        bx = torch.cat([rnd(9, 3, lo=-2, hi=2), rnd(9, 3, lo=.3, hi=1.5), rnd(9, 1, lo=-3.1, hi=3.1), rnd(9, 2, lo=-.3, hi=.3)], 1)
        boxes = EulerDepthInstance3DBoxes(bx.clone())
        p = pts.clone()
        if hf:
            p = boxes.flip('horizontal', points=p)
        if vf:
            p = boxes.flip('vertical', points=p)
        angle = -float(rnd(1, lo=-0.087266, hi=0.087266))             # rot_dof 1: noise_rotation *= -1
        p, rot_mat_T = boxes.rotate(angle, p)
        scale = float(rnd(1, lo=.9, hi=1.1))
        boxes.scale(scale)
        p = p * scale                                                   # BasePoints.scale
        trans = rnd(3, lo=-.3, hi=.3).numpy()
        boxes.translate(trans)
        p = p + torch.from_numpy(trans)                                 # BasePoints.translate
        np.savez_compressed(os.path.join(out_dir, name + '.npz'), points=pts.numpy(), boxes=bx.numpy(), hflip=hf, vflip=vf,
                            angle=np.float32(angle), rot_mat_T=rot_mat_T.numpy(), scale=np.float32(scale), trans=trans,
                            points_out=p.numpy(), boxes_out=boxes.tensor.numpy())
    print('golden vectors written to', out_dir)


if __name__ == '__main__':
    main()
