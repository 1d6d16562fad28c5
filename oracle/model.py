"""The mv-3ddet train-step forward (losses) on the CPU, PyTorch f32.  TEST ORACLE.

Functional restatement, driven by a reference-named state dict, of:
  * Det3DDataPreprocessor.preprocess_img        data_preprocessors/data_preprocessor.py:249-264
  * ConvertRGBDToPoints / points_img2cam (A1)   datasets/transforms/points.py:30-81, structures/bbox_3d/utils.py:335-368
  * AggregateMultiViewPoints (A3)               datasets/transforms/multiview.py:139-169
  * mmdet.ResNet(depth=50, base_channels=16)    configs/detection/mv-det3d_...py:24-34 (un-vendored; torchvision-style bottleneck, stride on the 3x3)
  * MinkResNet(depth=34)                        models/backbones/mink_resnet.py:32-140
  * batch_point_sample (A8)                     models/layers/fusion_layers/point_fusion.py:20-107,208-311
  * SparseFeatureFusionSingleStage3DDetector.extract_feat / loss
                                                models/detectors/sparse_featfusion_single_stage.py:86-243
  * FCAF3DHeadRotMat forward / loss             models/dense_heads/fcaf3d_head.py:993-1020,1091-1350
"""
import numpy as np
import torch
import torch.nn.functional as F
from . import coords as C
from . import rounding as R
from . import sparse as S
from . import geometry as G

EPS32 = float(torch.finfo(torch.float32).eps)


# ----------------------------------------------------------------------------- data side
def preprocess_img(img_u8, mean, std):
    """A18.  (..., 3, H, W) uint8 BGR -> f32 RGB normalised: x[[2,1,0]].float(); (x-mean)/std."""
    x = img_u8.flip(-3).float()
    m = torch.tensor(mean, dtype=torch.float32).view(3, 1, 1)
    s = torch.tensor(std, dtype=torch.float32).view(3, 1, 1)
    return (x - m) / s


def unproject_depth(depth, cam2img):
    """A1.  depth (H,W) f32, cam2img 4x4 -> (H*W,3) camera-frame points and the
    nonzero mask (points.py:44-52, utils.py:335-368)."""
    h, w = depth.shape
    us, vs = torch.meshgrid(torch.arange(w, dtype=torch.float32), torch.arange(h, dtype=torch.float32), indexing='xy')
    d = depth.reshape(-1, 1)
    xys = torch.stack([us.reshape(-1), vs.reshape(-1)], 1)
    unnormed = torch.cat([xys * d, d], 1)
    pad = torch.eye(4, dtype=torch.float32)
    pad[:cam2img.shape[0], :cam2img.shape[1]] = cam2img
    inv = torch.inverse(pad).transpose(0, 1)
    homo = torch.cat([unnormed, torch.ones((h * w, 1))], 1)
    pts = torch.mm(homo, inv)[:, :3]
    return pts, (depth.reshape(-1) != 0)


def aggregate_points(points_cam, global2cam):
    """A3.  multiview.py:151-158: solve(global2cam, [p;1])."""
    p = torch.cat([points_cam, points_cam.new_ones(points_cam.shape[0], 1)], 1)
    return torch.linalg.solve(global2cam, p.t()).t()[:, :3]


# ----------------------------------------------------------------------------- 2D backbone
def _bn2d_eval(x, sd, p):
    return F.batch_norm(x, sd[p + '.running_mean'], sd[p + '.running_var'], sd[p + '.weight'], sd[p + '.bias'],
                        False, 0.0, 1e-5)


def _conv2d(x, w, stride=1, padding=0):
    """bias-free conv2d through the oracle's operand-rounding switch (oracle/rounding.py)"""
    return R.op(lambda a, b: F.conv2d(a, b, None, stride, padding), x, w, w.shape[1], w.shape[0])


def resnet50_w16(x, sd, prefix='backbone.'):
    """mmdet.ResNet depth=50 base_channels=16, norm_eval, out_indices (0,1,2,3)."""
    x = F.conv2d(x, sd[prefix + 'conv1.weight'], None, 2, 3)
    x = F.relu(_bn2d_eval(x, sd, prefix + 'bn1'))
    x = R.act(F.max_pool2d(x, 3, 2, 1))         # R.act: activation STORAGE rounding of the bf16 specification (identity otherwise)
    outs = []
    for li, nblk in enumerate((3, 4, 6, 3)):
        for bi in range(nblk):
            p = f'{prefix}layer{li + 1}.{bi}.'
            stride = 2 if (bi == 0 and li > 0) else 1
            idt = x
            o = R.act(F.relu(_bn2d_eval(_conv2d(x, sd[p + 'conv1.weight']), sd, p + 'bn1')))
            o = R.act(F.relu(_bn2d_eval(_conv2d(o, sd[p + 'conv2.weight'], stride, 1), sd, p + 'bn2')))
            o = _bn2d_eval(_conv2d(o, sd[p + 'conv3.weight']), sd, p + 'bn3')
            if bi == 0:
                idt = R.act(_bn2d_eval(_conv2d(x, sd[p + 'downsample.0.weight'], stride), sd, p + 'downsample.1'))
            x = R.act(F.relu(o + idt))
        outs.append(x)
    return outs


# ----------------------------------------------------------------------------- 3D backbone
def _bn(x, sd, p, training=True):
    return S.batch_norm(x, sd[p + '.bn.weight'], sd[p + '.bn.bias'], sd[p + '.bn.running_mean'],
                        sd[p + '.bn.running_var'], training)


def mink_resnet34(x, sd, prefix='backbone_3d.', training=True, trace=None):
    """mink_resnet.py:122-140 with ME BasicBlock (conv3-BN-ReLU-conv3-BN (+down) ReLU)."""
    x = S.conv(x, sd[prefix + 'conv1.kernel'], 3, 2)
    x = S.instance_norm(x, sd[prefix + 'norm1.weight'], sd[prefix + 'norm1.bias'])
    x = x.new(F.relu(x.feats))
    x = S.max_pool(x)
    outs = []
    for li, nblk in enumerate((3, 4, 6, 3)):
        for bi in range(nblk):
            p = f'{prefix}layer{li + 1}.{bi}.'
            stride = 2 if bi == 0 else 1
            o = S.conv(x, sd[p + 'conv1.kernel'], 3, stride)
            tr_c1 = o.feats
            o = _bn(o, sd, p + 'norm1', training)
            o = o.new(F.relu(o.feats))
            tr_n1 = o.feats
            o = S.conv(o, sd[p + 'conv2.kernel'], 3, 1)
            tr_c2 = o.feats
            o = _bn(o, sd, p + 'norm2', training)
            if bi == 0:
                idt = S.conv(x, sd[p + 'downsample.0.kernel'], 1, stride)
                idt = _bn(idt, sd, p + 'downsample.1', training)
            else:
                idt = x
            x = o.new(F.relu(o.feats + idt.feats))
            if trace is not None:
                for nm, t in (('conv1', tr_c1), ('norm1', tr_n1), ('conv2', tr_c2), ('out', x.feats)):
                    t.retain_grad()
                    trace.append((f'layer{li + 1}.{bi}.{nm}', t))
        outs.append(x)
    return outs


# ----------------------------------------------------------------------------- fusion (A8)
def apply_3d_transformation_reverse(pcd, meta):
    """point_fusion.py:20-107 with reverse=True, DEPTH points (flip: depth_points.py:39-50)."""
    pcd = pcd.clone()
    rot = torch.tensor(meta['pcd_rotation'], dtype=pcd.dtype) if 'pcd_rotation' in meta else torch.eye(3, dtype=pcd.dtype)
    scale = meta.get('pcd_scale_factor', 1.)
    trans = torch.tensor(meta['pcd_trans'], dtype=pcd.dtype) if 'pcd_trans' in meta else torch.zeros(3, dtype=pcd.dtype)
    hf, vf = meta.get('pcd_horizontal_flip', False), meta.get('pcd_vertical_flip', False)
    flow = meta.get('transformation_3d_flow', [])
    for op in flow[::-1]:
        if op == 'T':
            pcd[:, :3] += (-trans)
        elif op == 'S':
            pcd[:, :3] *= (1.0 / scale)
        elif op == 'R':
            pcd[:, :3] = pcd[:, :3] @ rot.inverse()
        elif op == 'HF':
            if hf:
                pcd[:, 0] = -pcd[:, 0]
        elif op == 'VF':
            if vf:
                pcd[:, 1] = -pcd[:, 1]
        else:
            raise AssertionError(op)
    return pcd


def batch_point_sample(meta, img_features, points, proj_mat, img_scale_factor, img_crop_offset, img_flip,
                       img_pad_shape, img_shape):
    """point_fusion.py:208-311 with aligned=False, padding 'zeros', align_corners=True,
    valid_flag=True.  img_features (V,C,H,W); points (N,3); proj_mat (V,4,4)."""
    points = apply_3d_transformation_reverse(points, meta)
    proj_mat = proj_mat.to(points.dtype)
    points = points.repeat(proj_mat.shape[0], 1, 1)
    p4 = torch.cat([points, points.new_ones(points.shape[:-1] + (1,))], dim=-1)
    p2 = torch.bmm(p4, proj_mat.permute(0, 2, 1))
    pts_2d = p2[..., :2] / p2[..., 2:3].clamp(min=1e-3)
    depths = p2[..., 2]
    img_coors = pts_2d * img_scale_factor
    img_coors = img_coors - img_crop_offset
    coor_x, coor_y = torch.split(img_coors, 1, dim=2)
    if img_flip:
        coor_x = img_shape[1] - coor_x
    h, w = img_pad_shape
    grid = torch.cat([coor_x / w * 2 - 1, coor_y / h * 2 - 1], dim=2).unsqueeze(1)
    pf = F.grid_sample(img_features, grid, mode='nearest', padding_mode='zeros', align_corners=True)
    valid = (coor_x.squeeze(2) < w) & (coor_x.squeeze(2) > 0) & (coor_y.squeeze(2) < h) & \
        (coor_y.squeeze(2) > 0) & (depths > 0)
    valid_num = valid.sum(dim=0)
    feats = pf.squeeze(2).sum(dim=0).t()
    feats = torch.where((valid_num > 0)[:, None], feats, torch.zeros_like(feats))
    return feats / torch.clamp(valid_num[:, None], min=1)


def projection_matrices(meta):
    """sparse_featfusion_single_stage.py:160-164: intrinsic @ extrinsic per view (f32)."""
    d2i = meta['depth2img']
    mats = []
    for i in range(len(d2i['extrinsic'])):
        intr = torch.tensor(np.asarray(d2i['intrinsic'][i]), dtype=torch.float32)
        extr = torch.tensor(np.asarray(d2i['extrinsic'][i]), dtype=torch.float32)
        mats.append(intr @ extr)
    return torch.stack(mats)


# ----------------------------------------------------------------------------- detector
def extract_feat(sd, points, imgs, metas, voxel_size=0.01, training=True, trace=None):
    """sparse_featfusion_single_stage.py:86-221 (use_xyz_feat=True)."""
    n_batch = len(points)
    coords, src = C.voxelize([p.detach().numpy() for p in points], voxel_size)
    feats = torch.cat([p[:, :3] for p in points])[torch.from_numpy(src)]
    x = S.SpT(coords, feats, 1, n_batch, {})
    xs = mink_resnet34(x, sd, training=training, trace=trace)
    B, V = imgs.shape[:2]
    img_feats = resnet50_w16(imgs.reshape((-1,) + imgs.shape[2:]), sd)
    img_feats = [f.reshape((B, V) + f.shape[1:]) for f in img_feats]
    outs = []
    for lvl, xl in enumerate(xs):
        per_sample = []
        for b in range(n_batch):
            meta = metas[b]
            rows = xl.batch_rows(b)
            pts = (torch.from_numpy(xl.coords[rows, 1:]).float() * voxel_size).to(xl.feats.dtype)
            sf = torch.tensor(meta['scale_factor'][:2], dtype=torch.float32) if 'scale_factor' in meta else 1
            off = torch.tensor(meta['img_crop_offset'], dtype=torch.float32) if 'img_crop_offset' in meta else 0
            per_sample.append(batch_point_sample(meta, img_feats[lvl][b], pts, projection_matrices(meta), sf, off,
                                                 meta.get('flip', False), imgs.shape[-2:], meta['img_shape'][:2]))
        outs.append(xl.new(torch.cat([xl.feats, torch.cat(per_sample)], 1)))
    return outs


def _block(x, sd, p, training):
    x = S.conv(x, sd[p + '.0.kernel'], 3)
    x = _bn(x, sd, p + '.1', training)
    return x.new(F.elu(x.feats))


def _up_block(x, sd, p, training):
    x = S.gen_conv_transpose(x, sd[p + '.0.kernel'])
    x = _bn(x, sd, p + '.1', training)
    x = x.new(F.elu(x.feats))
    x = S.conv(x, sd[p + '.3.kernel'], 3)
    x = _bn(x, sd, p + '.4', training)
    return x.new(F.elu(x.feats))


def prune_mask(x, scores, thr):
    """fcaf3d_head.py:1091-1114.  Deterministic tie rule for topk(sorted=False):
    keep the `thr` largest interpolated scores, ties broken by lower row first."""
    with torch.no_grad():
        s = S.features_at_coordinates(scores, x.coords)[:, 0]
        mask = np.zeros(x.coords.shape[0], bool)
        for b in range(x.n_batch):
            rows = x.batch_rows(b)
            k = min(len(rows), thr)
            if k == len(rows):
                mask[rows] = True
                continue
            sb = s[torch.from_numpy(rows)]
            order = torch.argsort(sb, descending=True, stable=True)[:k]
            mask[rows[order.numpy()]] = True
    return mask


def head_forward(xs, sd, prefix='bbox_head.', voxel_size=0.01, thr=100000, training=True, trace=None):
    """fcaf3d_head.py:993-1020,1116-1149.  Returns per-level lists (fine->coarse) of
    per-sample tensors: center (N,1), bbox (N,12), cls (N,C), points (N,3)."""
    n_lvl = len(xs)
    outs = [None] * n_lvl
    x = xs[-1]
    score = None
    for i in range(n_lvl - 1, -1, -1):
        if i < n_lvl - 1:
            x = _up_block(x, sd, f'{prefix}up_block_{i + 1}', training)
            x = S.union_add(x, xs[i])          # rows: generated children first, then the backbone voxels they miss (our row-order spec, round 6)
            x = S.prune(x, prune_mask(x, score, thr))
        out = _block(x, sd, f'{prefix}out_block_{i}', training)
        mm = lambda w: R.op(lambda a, b: a @ b, out.feats, w, out.feats.shape[1], 16)    # one padded 320-column GEMM on the device
        center = mm(sd[prefix + 'conv_center.kernel'])
        cls = mm(sd[prefix + 'conv_cls.kernel']) + sd[prefix + 'conv_cls.bias']
        reg = mm(sd[prefix + 'conv_reg.kernel'])
        dist = torch.exp(reg[:, :6] * sd[f'{prefix}scales.{i}.scale']).clamp(min=1e-3)
        bbox = torch.cat((dist, reg[:, 6:]), 1)
        if trace is not None:
            for nm, t in (('out', out.feats), ('center', center), ('reg', reg), ('cls', cls), ('x', x.feats)):
                if t.requires_grad:
                    t.retain_grad()
                trace.append((f'head.L{i}.{nm}', t))
        score = out.new(cls.detach().max(dim=1, keepdim=True).values)
        per = []
        for b in range(out.n_batch):
            r = torch.from_numpy(out.batch_rows(b))
            per.append((center[r], bbox[r], cls[r], torch.from_numpy(out.coords[r.numpy(), 1:]).float() * voxel_size))
        outs[i] = per
    return outs


def loss_single(level_preds, gt_boxes, gt_labels, world_size_mean=lambda t: t,
                decouple_weights=(0.2, 0.2, 0.2, 0.4), targets_override=None):
    """fcaf3d_head.py:1151-1294 for one sample.  level_preds: list over levels of
    (center, bbox, cls, points)."""
    points_l = [p[3] for p in level_preds]
    if targets_override is None:
        center_t, bbox_t, cls_t = G.get_targets(points_l, gt_boxes, gt_labels)
    else:   # f64 "truth" runs reuse the f32 integer decisions
        center_t, bbox_t, cls_t = (targets_override[0].to(level_preds[0][0].dtype),
                                   targets_override[1].to(level_preds[0][0].dtype), targets_override[2])
    center_p = torch.cat([p[0] for p in level_preds])
    bbox_p = torch.cat([p[1] for p in level_preds])
    cls_p = torch.cat([p[2] for p in level_preds])
    points = torch.cat(points_l)
    pos = torch.nonzero(cls_t >= 0).squeeze(1)
    n_pos = max(float(world_size_mean(torch.tensor(float(len(pos))))), 1.)
    cls_loss = G.sigmoid_focal_loss_sum(cls_p, cls_t) / (n_pos + EPS32)
    if len(pos) > 0:
        ct = center_t[pos].unsqueeze(1)
        center_loss = F.binary_cross_entropy_with_logits(center_p[pos], ct, reduction='none').sum() / (n_pos + EPS32)
        tb = bbox_t[pos]
        dec = G.bbox_pred_to_bbox(points[pos], bbox_p[pos])
        w = decouple_weights
        bbox_loss = w[0] * G.bbox_cd_loss(torch.cat((dec[:, :3], tb[:, 3:6], tb[:, 6:]), -1), tb)
        bbox_loss = bbox_loss + w[1] * G.bbox_cd_loss(torch.cat((tb[:, :3], dec[:, 3:6], tb[:, 6:]), -1), tb)
        bbox_loss = bbox_loss + w[2] * G.bbox_cd_loss(torch.cat((tb[:, :3], tb[:, 3:6], dec[:, 6:]), -1), tb)
        bbox_loss = bbox_loss + w[3] * G.bbox_cd_loss(dec, tb)
    else:
        center_loss = center_p[pos].sum()
        bbox_loss = bbox_p[pos].sum()
    return center_loss, bbox_loss, cls_loss, (center_t, bbox_t, cls_t)


def detector_loss(sd, points, imgs, metas, gt_boxes, gt_labels, voxel_size=0.01, thr=100000, training=True,
                  return_aux=False, targets_override=None, trace=None):
    """SparseFeatureFusionSingleStage3DDetector.loss -> dict(loss_center, loss_bbox, loss_cls)."""
    xs = extract_feat(sd, points, imgs, metas, voxel_size, training, trace)
    outs = head_forward(xs, sd, voxel_size=voxel_size, thr=thr, training=training, trace=trace)
    n_batch = len(points)
    cl, bl, kl, aux = [], [], [], []
    for b in range(n_batch):
        c, bb, k, tg = loss_single([outs[l][b] for l in range(len(outs))], gt_boxes[b], gt_labels[b],
                                   targets_override=None if targets_override is None else targets_override[b])
        cl.append(c), bl.append(bb), kl.append(k), aux.append(tg)
    losses = dict(loss_center=torch.stack(cl).mean(), loss_bbox=torch.stack(bl).mean(), loss_cls=torch.stack(kl).mean())
    if return_aux:
        return losses, dict(xs=xs, outs=outs, targets=aux)
    return losses


def detector_predict(sd, points, imgs, metas, voxel_size=0.01, thr=100000, nms_pre=1000, score_thr=0.01, iou_thr=0.5):
    """SparseFeatureFusionSingleStage3DDetector.predict (eval mode): list over samples of (boxes (M,9), scores, labels)."""
    from . import predict as PR
    with torch.no_grad():
        xs = extract_feat(sd, points, imgs, metas, voxel_size, training=False)
        outs = head_forward(xs, sd, voxel_size=voxel_size, thr=thr, training=False)
        return [PR.predict_single([outs[l][b] for l in range(len(outs))], nms_pre, score_thr, iou_thr)
                for b in range(len(points))]
