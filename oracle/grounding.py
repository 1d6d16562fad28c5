"""The grounding path (BASELINE config 4) on the CPU, PyTorch f32 / numpy f64.  TEST ORACLE.

Functional restatement, driven by a reference-named state dict, of:
  * MinkNeck.forward / _prune / convert_to_batch          models/necks/mink_neck.py:133-244
  * SparseFeatureFusion3DGrounder.pre_decoder / forward_decoder   models/detectors/sparse_featfusion_grounder.py:324-447
  * PositionEmbeddingLearned, SparseFeatureFusionTransformerDecoderLayer / Decoder
                                                          models/layers/ground_transformer/decoder.py:20-297
      (mmcv MultiheadAttention / FFN are un-vendored: identity + nn.MultiheadAttention(q + q_pos, k + k_pos, v),
       x + Linear(ReLU(Linear(x))))
  * ContrastiveEmbed, GroundingHead._bbox_pred_to_bbox ('baseline', 9), _get_targets_single, loss_by_feat_single
                                                          models/dense_heads/grounding_head.py:20-99,267-296,365-425,686-822
  * HungarianAssigner3D.assign                            models/task_modules/assigners/hungarian_assigner.py:56-138
  * BinaryFocalLossCost, BBox3DL1Cost, IoU3DCost          models/losses/match_cost.py:49-75,95-113,213-265
  * EulerInstance3DBoxes.overlaps -> pytorch3d box3d_overlap (un-vendored): exact polyhedral IoU, restated twice:
      `box3d_iou_qhull` (scipy half-space intersection + convex hull: the independent anchor) and `box3d_iou` (face
      clipping, numpy f64, what the model-level oracle uses); tests/test_oracle_golden.py pins one against the other.
Pinned by tests/golden/ground_*.npz recorded from the reference's own classes (oracle/make_golden_ground.py)."""
import math
import numpy as np
import torch
import torch.nn.functional as F
from . import coords as C
from . import geometry as G
from . import model as M
from . import rounding as R
from . import sparse as S

EPS32 = float(torch.finfo(torch.float32).eps)


# ----------------------------------------------------------------------------- exact IoU of oriented boxes
def _box_frame(b):
    b = np.asarray(b, np.float64)
    R = G.euler_to_matrix_zxy(torch.tensor(b[6:9], dtype=torch.float64)).numpy()
    return b[:3], R, b[3:6] / 2          # centre, axes = columns of R, half sizes


def _halfspaces(b):
    c, R, h = _box_frame(b)
    hs = []
    for j in range(3):
        for s in (1.0, -1.0):
            n = s * R[:, j]
            hs.append(np.concatenate([n, [-(n @ c) - h[j]]]))      # n.x + d <= 0
    return np.array(hs)


def box3d_iou_qhull(a, b):
    """independent anchor: Chebyshev centre (LP) -> scipy HalfspaceIntersection -> ConvexHull volume"""
    from scipy.optimize import linprog
    from scipy.spatial import ConvexHull, HalfspaceIntersection
    hs = np.concatenate([_halfspaces(a), _halfspaces(b)])
    A, d = hs[:, :3], hs[:, 3]
    res = linprog(c=[0, 0, 0, -1], A_ub=np.hstack([A, np.ones((12, 1))]), b_ub=-d, bounds=[(None, None)] * 3 + [(0, None)])
    va, vb = float(np.prod(np.asarray(a)[3:6])), float(np.prod(np.asarray(b)[3:6]))
    if not res.success or res.x[3] <= 1e-9:
        return 0.0
    vol = ConvexHull(HalfspaceIntersection(hs, res.x[:3]).intersections).volume
    return vol / (va + vb - vol)


def _clip_faces(P, Q, O, inclusive):
    cP, RP, hP = P
    cQ, RQ, hQ = Q
    total = 0.0
    for j in range(3):
        k, l = (j + 1) % 3, (j + 2) % 3
        for sgn in (1.0, -1.0):
            n = sgn * RP[:, j]
            fc = cP + n * hP[j]
            ek, el = RP[:, k] * hP[k], RP[:, l] * hP[l]
            poly = [fc + ek + el, fc - ek + el, fc - ek - el, fc + ek - el]
            for jj in range(3):
                for sg in (1.0, -1.0):
                    m = sg * RQ[:, jj]
                    same = inclusive and (m @ n) > 0.5
                    out = []
                    for i in range(len(poly)):
                        p, q = poly[i], poly[(i + 1) % len(poly)]
                        dp, dq = m @ (p - cQ) - hQ[jj], m @ (q - cQ) - hQ[jj]
                        ip = dp < -1e-12 or (same and dp <= 1e-12)
                        iq = dq < -1e-12 or (same and dq <= 1e-12)
                        if ip:
                            out.append(p)
                        if ip != iq:
                            out.append(p + (q - p) * (dp / (dp - dq)))
                    poly = out
                    if not poly:
                        break
                if not poly:
                    break
            if len(poly) < 3:
                continue
            av = np.zeros(3)
            for i in range(1, len(poly) - 1):
                av += np.cross(poly[i] - poly[0], poly[i + 1] - poly[0])
            total += (n @ (fc - O)) * 0.5 * np.linalg.norm(av)
    return total


def box3d_iou(a, b):
    """intersection polytope by clipping each face of one box with the half-spaces of the other; volume from the
    divergence theorem (f64)"""
    A, B = _box_frame(a), _box_frame(b)
    va, vb = float(np.prod(np.asarray(a, np.float64)[3:6])), float(np.prod(np.asarray(b, np.float64)[3:6]))
    if np.sum((A[0] - B[0]) ** 2) > (np.linalg.norm(A[2]) + np.linalg.norm(B[2])) ** 2:
        return 0.0
    v = max((_clip_faces(A, B, A[0], True) + _clip_faces(B, A, A[0], False)) / 3.0, 0.0)
    return v / (va + vb - v)


def overlaps(boxes1, boxes2):
    """EulerInstance3DBoxes.overlaps: (N,9) x (M,9) -> (N,M) f32 IoU"""
    out = torch.zeros((boxes1.shape[0], boxes2.shape[0]), dtype=torch.float32)
    for i in range(boxes1.shape[0]):
        for j in range(boxes2.shape[0]):
            out[i, j] = box3d_iou(boxes1[i].detach().numpy(), boxes2[j].detach().numpy())
    return out


# ----------------------------------------------------------------------------- match costs + assignment
def binary_focal_cost(scores, positive_maps, text_token_mask, alpha=0.25, gamma=2, eps=1e-12, weight=1.0):
    idx = torch.nonzero(text_token_mask[0]).squeeze(-1)
    p = scores[:, idx].flatten(1).sigmoid()
    y = positive_maps[:, idx].flatten(1).float()
    neg = -(1 - p + eps).log() * (1 - alpha) * p.pow(gamma)
    pos = -(p + eps).log() * alpha * (1 - p).pow(gamma)
    return (torch.einsum('nc,mc->nm', pos, y) + torch.einsum('nc,mc->nm', neg, (1 - y))) * weight


def hungarian_assign(scores, boxes, gt_boxes, positive_maps, text_token_mask, w_cls=1.0, w_l1=2.0, w_iou=2.0, return_cost=False):
    """-> gt_inds (Q,) long: 0 background, k > 0 matched to gt k-1"""
    from scipy.optimize import linear_sum_assignment
    Q, Gn = boxes.shape[0], gt_boxes.shape[0]
    gt_inds = torch.zeros(Q, dtype=torch.long)
    if Gn == 0 or Q == 0:
        return (gt_inds, None) if return_cost else gt_inds
    cost = binary_focal_cost(scores, positive_maps, text_token_mask, weight=w_cls) + torch.cdist(boxes, gt_boxes, p=1) * w_l1 + \
        -overlaps(boxes, gt_boxes) * w_iou
    cost = torch.nan_to_num(cost.detach(), nan=100.0, posinf=100.0, neginf=-100.0)
    r, c = linear_sum_assignment(cost)
    gt_inds[torch.from_numpy(r)] = torch.from_numpy(c) + 1
    return (gt_inds, cost) if return_cost else gt_inds


# ----------------------------------------------------------------------------- head
def contrastive_embed(visual, text, text_token_mask, bias, visual_mask=None, max_text_len=256):
    res = visual @ text.transpose(-1, -2) / math.sqrt(visual.shape[-1]) + bias
    res = res.masked_fill(~text_token_mask[:, None, :], float('-inf'))
    if visual_mask is not None:
        res = res.masked_fill(~visual_mask[:, :, None], float('-inf'))
    new = torch.full((*res.shape[:-1], max_text_len), float('-inf'))
    new[..., :res.shape[-1]] = res
    return new


def reg_branch(x, sd, p='bbox_head.reg_branches.0.'):
    h = F.relu(F.linear(x, sd[p + '0.weight'], sd[p + '0.bias']))
    h = F.relu(F.linear(h, sd[p + '2.weight'], sd[p + '2.bias']))
    return F.linear(h, sd[p + '4.weight'], sd[p + '4.bias'])


def bbox_pred_to_bbox(points, pred, coder='baseline'):
    """GroundingHead._bbox_pred_to_bbox with 9 outputs: 'baseline' (grounding_head.py:292-296) = (offset + point, clamped exp of
    the log size, Euler angles); 'FCAF' (:308-363, configs/grounding/..._fcaf-coder.py) = the six outputs are log distances to
    the faces (exp, clamp 2e-2 -- written in place by the reference, a pure function for autograd), the centre is the point
    shifted by the ROTATED half difference of opposite distances (rotation_3d_in_euler: shift @ R^T, R = Rz Rx Ry), the size
    is the sum of opposite distances"""
    if coder == 'baseline':
        return torch.cat((pred[..., :3] + points, torch.exp(pred[..., 3:6]).clamp(min=2e-2), pred[..., 6:]), -1)
    assert coder == 'FCAF' and pred.shape[-1] == 9
    d = torch.exp(pred[..., :6]).clamp(min=2e-2)
    shift = torch.stack(((d[..., 1] - d[..., 0]) / 2, (d[..., 3] - d[..., 2]) / 2, (d[..., 5] - d[..., 4]) / 2), -1)
    R = G.euler_to_matrix_zxy(pred[..., 6:9])
    center = points + torch.matmul(R, shift.unsqueeze(-1)).squeeze(-1)
    size = torch.stack((d[..., 0] + d[..., 1], d[..., 2] + d[..., 3], d[..., 4] + d[..., 5]), -1)
    return torch.cat((center, size, pred[..., 6:9]), -1)


def py_sigmoid_focal_loss_sum(pred, target, gamma=2.0, alpha=0.25):
    p = pred.sigmoid()
    pt = (1 - p) * target + p * (1 - target)
    fw = (alpha * target + (1 - alpha) * (1 - target)) * pt.pow(gamma)
    return (F.binary_cross_entropy_with_logits(pred, target, reduction='none') * fw).sum()


def loss_by_feat_single(cls_scores, pred_bboxes, gt_boxes_list, positive_maps_list, text_token_mask, weights=(0.2, 0.2, 0.2, 0.4),
                        world_mean=lambda t: t, return_assign=False):
    """grounding_head.py:686-822 for one decoder layer.  cls_scores (B,Q,256) (-inf padded), pred_bboxes (B,Q,9)."""
    B, Q = cls_scores.shape[:2]
    labels = torch.zeros_like(cls_scores)
    tgt = torch.zeros_like(pred_bboxes)
    wgt = torch.zeros(B, Q)
    assigns, n_pos = [], 0
    with torch.no_grad():
        for b in range(B):
            tm = text_token_mask[b][None].repeat(max(len(gt_boxes_list[b]), 1), 1)
            gi = hungarian_assign(cls_scores[b], pred_bboxes[b], gt_boxes_list[b], positive_maps_list[b], tm)
            assigns.append(gi)
            pos = torch.nonzero(gi > 0).squeeze(-1)
            labels[b, pos] = positive_maps_list[b][gi[pos] - 1]
            tgt[b, pos] = gt_boxes_list[b][gi[pos] - 1]
            wgt[b, pos] = 1.0
            n_pos += len(pos)
    tmask = torch.zeros((B, cls_scores.shape[-1]), dtype=torch.bool)
    tmask[:, :text_token_mask.shape[1]] = text_token_mask
    sel = tmask[:, None, :].repeat(1, Q, 1)
    avg = max(float(world_mean(torch.tensor([float(n_pos)]))), 1.0)
    loss_cls = py_sigmoid_focal_loss_sum(cls_scores[sel], labels[sel]) / (avg + EPS32)
    vp, vt = pred_bboxes.reshape(-1, 9)[wgt.reshape(-1) > 0], tgt.reshape(-1, 9)[wgt.reshape(-1) > 0]
    w = weights
    lb = w[0] * G.bbox_cd_loss(torch.cat((vp[:, :3], vt[:, 3:6], vt[:, 6:]), -1), vt)
    lb = lb + w[1] * G.bbox_cd_loss(torch.cat((vt[:, :3], vp[:, 3:6], vt[:, 6:]), -1), vt)
    lb = lb + w[2] * G.bbox_cd_loss(torch.cat((vt[:, :3], vt[:, 3:6], vp[:, 6:]), -1), vt)
    lb = lb + w[3] * G.bbox_cd_loss(vp, vt)
    return (loss_cls, lb, assigns) if return_assign else (loss_cls, lb)


# ----------------------------------------------------------------------------- neck
def _block(x, sd, p, training):
    x = S.conv(x, sd[p + '.0.kernel'], 3)
    x = M._bn(x, sd, p + '.1', training)
    return x.new(F.elu(x.feats))


def mink_neck(xs, sd, batch_size, prefix='neck_3d.', voxel_size=0.01, thr=1000, training=True):
    """mink_neck.py:133-244 -> per-sample lists (feats, scores, points), levels concatenated coarse -> fine"""
    n_lvl = len(xs)
    feats, cls_preds, points = [], [], []
    x = xs[-1]
    score = None
    for i in range(n_lvl - 1, -1, -1):
        if i < n_lvl - 1:
            x = M._up_block(x, sd, f'{prefix}up_block_{i + 1}', training)
            x = S.union_add(x, xs[i])          # rows: generated children first, then the backbone voxels they miss (our row-order spec, round 6)
            x = S.prune(x, M.prune_mask(x, score, thr))
        out = _block(x, sd, f'{prefix}out_block_{i}', training)
        cls = R.op(lambda a, b: a @ b, out.feats.detach(), sd[prefix + 'conv_cls.kernel'], out.feats.shape[1]) + sd[prefix + 'conv_cls.bias']
        score = out.new(cls.max(dim=1, keepdim=True).values)
        rows = [torch.from_numpy(out.batch_rows(b)) for b in range(batch_size)]
        feats.append([out.feats[r] for r in rows])
        cls_preds.append([cls[r] for r in rows])
        points.append([torch.from_numpy(out.coords[r.numpy(), 1:]).float() * voxel_size for r in rows])
    cat = lambda L: [torch.cat([lvl[b] for lvl in L], 0) for b in range(batch_size)]
    return cat(feats), cat(cls_preds), cat(points)


# ----------------------------------------------------------------------------- decoder
def _conv1d(x, w, b):
    """1x1 Conv1d (a Linear layer over positions) through the operand-rounding switch"""
    return R.op(lambda a, c: F.conv1d(a, c), x, w, w.shape[1], w.shape[0]) + b[None, :, None]


def posembed(xyz, sd, p, training=True):
    """PositionEmbeddingLearned: (B, N, c) -> (B, N, E); BatchNorm1d over all B*N positions"""
    q = p + '.position_embedding_head'
    x = xyz.transpose(1, 2).contiguous()
    x = _conv1d(x, sd[q + '.0.weight'], sd[q + '.0.bias'])
    x = F.relu(F.batch_norm(x, sd[q + '.1.running_mean'], sd[q + '.1.running_var'], sd[q + '.1.weight'], sd[q + '.1.bias'],
                            training, 0.1, 1e-5))
    x = _conv1d(x, sd[q + '.3.weight'], sd[q + '.3.bias'])
    return x.transpose(1, 2).contiguous()


def _mha(query, key, value, sd, p, H, query_pos=None, key_pos=None, key_padding_mask=None):
    """mmcv MultiheadAttention(batch_first=True, dropout 0): identity + nn.MultiheadAttention(q + q_pos, k + k_pos, v)"""
    if key_pos is None and query_pos is not None and query_pos.shape == key.shape:
        key_pos = query_pos
    q = query + query_pos if query_pos is not None else query
    k = key + key_pos if key_pos is not None else key
    out = F.multi_head_attention_forward(q.transpose(0, 1), k.transpose(0, 1), value.transpose(0, 1), q.shape[-1], H,
                                         sd[p + '.attn.in_proj_weight'], sd[p + '.attn.in_proj_bias'], None, None, False, 0.0,
                                         sd[p + '.attn.out_proj.weight'], sd[p + '.attn.out_proj.bias'], training=False,
                                         key_padding_mask=key_padding_mask, need_weights=False)[0]
    return query + out.transpose(0, 1)


def _ln(x, sd, p):
    return F.layer_norm(x, (x.shape[-1],), sd[p + '.weight'], sd[p + '.bias'], 1e-5)


def decoder_layer(query, key, value, query_pos, key_pos, kpm, text, tpm, sd, p, H=8):
    query = _ln(_mha(query, query, query, sd, p + 'self_attn', H, query_pos, query_pos), sd, p + 'norms.0')
    query = _ln(_mha(query, text, text, sd, p + 'cross_attn_text', H, query_pos, None, tpm), sd, p + 'norms.1')
    query = _ln(_mha(query, key, value, sd, p + 'cross_attn', H, query_pos, key_pos, kpm), sd, p + 'norms.2')
    h = F.linear(F.relu(F.linear(query, sd[p + 'ffn.layers.0.0.weight'], sd[p + 'ffn.layers.0.0.bias'])),
                 sd[p + 'ffn.layers.1.weight'], sd[p + 'ffn.layers.1.bias'])
    return _ln(query + h, sd, p + 'norms.3')


def forward_transformer(feats_list, xyz_list, text_feats, text_token_mask, sd, num_queries=256, num_layers=6, H=8, training=True,
                        coder='baseline'):
    """pre_decoder + forward_decoder (sparse_featfusion_grounder.py:324-447) -> (hidden (L,B,Q,E), boxes (L,B,Q,9), aux)"""
    B = len(feats_list)
    Lmax, Lmin = max(f.shape[0] for f in feats_list), min(f.shape[0] for f in feats_list)
    E = feats_list[0].shape[1]
    feats = torch.stack([torch.cat([f, f.new_zeros(Lmax - f.shape[0], E)]) for f in feats_list])
    coords = torch.stack([torch.cat([c, c.new_zeros(Lmax - c.shape[0], 3)]) for c in xyz_list])
    fmask = torch.stack([torch.arange(Lmax) < f.shape[0] for f in feats_list])
    bias = sd['bbox_head.cls_branches.0.bias']
    enc = contrastive_embed(feats, text_feats, text_token_mask, bias, fmask)
    topk = min(num_queries, Lmin)
    sc = enc.max(-1)[0]
    # torch.topk leaves the order among equal scores unspecified: descending score, ties by lower index
    idx = torch.stack([torch.argsort(sc[b], descending=True, stable=True)[:topk] for b in range(B)])
    boxes0 = bbox_pred_to_bbox(coords, reg_branch(feats, sd), coder)
    g3, g9, gE = idx.unsqueeze(-1).repeat(1, 1, 3), idx.unsqueeze(-1).repeat(1, 1, 9), idx.unsqueeze(-1).repeat(1, 1, E)
    qcoords, pred, query = torch.gather(coords, 1, g3), torch.gather(boxes0, 1, g9).detach().clone(), torch.gather(feats, 1, gE)
    inter, inter_boxes = [], []
    for lid in range(num_layers):
        query_pos = posembed(pred, sd, 'decoder.self_posembed', training)
        key_pos = posembed(coords, sd, 'decoder.cross_posembed', training)
        query = decoder_layer(query, feats, feats, query_pos, key_pos, ~fmask, text_feats, ~text_token_mask, sd,
                              f'decoder.layers.{lid}.', H)
        new = bbox_pred_to_bbox(qcoords, reg_branch(query, sd), coder)
        pred = new.detach().clone()
        inter.append(_ln(query, sd, 'decoder.norm'))
        inter_boxes.append(new)
    return torch.stack(inter), torch.stack(inter_boxes), dict(idx=idx, scores=sc, pred0=torch.gather(boxes0, 1, g9), feats=feats,
                                                              coords=coords)


def head_loss(hidden, boxes, text_feats, text_token_mask, sd, gt_boxes_list, positive_maps_list, weights=(0.2, 0.2, 0.2, 0.4),
              return_aux=False):
    """GroundingHead.loss (grounding_head.py:606-684)"""
    L = hidden.shape[0]
    out, aux = {}, []
    for l in range(L):
        cls = contrastive_embed(hidden[l], text_feats, text_token_mask, sd['bbox_head.cls_branches.0.bias'])
        lc, lb, asg = loss_by_feat_single(cls, boxes[l], gt_boxes_list, positive_maps_list, text_token_mask, weights, return_assign=True)
        name = '' if l == L - 1 else f'd{l}.'
        out[name + 'loss_cls'], out[name + 'loss_bbox'] = lc, lb
        aux.append(dict(cls=cls, assign=asg))
    return (out, aux) if return_aux else out


def grounder_loss(sd, points, imgs, metas, text_hidden, text_token_mask, gt_boxes_list, positive_maps_list, num_queries=256,
                  num_layers=6, voxel_size=0.01, thr=1000, return_aux=False, coder='baseline'):
    """SparseFeatureFusion3DGrounder.loss with the frozen text encoder's output `text_hidden` (B,T,D) as an input"""
    xs = M.extract_feat(sd, points, imgs, metas, voxel_size, True)
    fl, sl, pl = mink_neck(xs, sd, len(points), voxel_size=voxel_size, thr=thr)
    text = F.linear(text_hidden, sd['text_feat_map.weight'], sd['text_feat_map.bias'])
    hidden, boxes, aux = forward_transformer(fl, pl, text, text_token_mask, sd, num_queries, num_layers, coder=coder)
    losses, haux = head_loss(hidden, boxes, text, text_token_mask, sd, gt_boxes_list, positive_maps_list, return_aux=True)
    if return_aux:
        aux.update(hidden=hidden, boxes=boxes, head=haux, text=text, feats_list=fl, points_list=pl)
        return losses, aux
    return losses
