"""Inference post-processing on the CPU.  TEST ORACLE.

Restates
  * FCAF3DHeadRotMat._predict_by_feat_single / _single_scene_multiclass_nms
        embodiedscan/models/dense_heads/fcaf3d_head.py:1352-1399,1666-1725
  * mmcv.ops.nms3d (un-vendored, mmcv 2.0.0rc4: iou3d_nms3d_forward = greedy NMS on the rotated BEV IoU of
    (x, y, z, dx, dy, dz, heading) boxes; kernel from OpenPCDet's iou3d_nms) -- restated from the published
    algorithm: rotated-rectangle intersection by edge crossings + contained corners, fan triangulation, EPS = 1e-8,
    in-box MARGIN = 1e-2, evaluated in float64 (mmcv: float32).  "Parity unpinned" (no mmcv here); the HIP kernel is held
    to THIS restatement and also evaluates the overlap in float64.
"""
import math
import torch
from . import geometry as G

EPS = 1e-8
MARGIN = 1e-2


def _corners(box):
    x, y, dx, dy, ang = float(box[0]), float(box[1]), float(box[3]), float(box[4]), float(box[6])
    c, s = math.cos(ang), math.sin(ang)
    x1, y1, x2, y2 = x - dx / 2, y - dy / 2, x + dx / 2, y + dy / 2
    pts = []
    for px, py in ((x1, y1), (x2, y1), (x2, y2), (x1, y2)):
        nx = (px - x) * c + (py - y) * (-s) + x
        ny = (px - x) * s + (py - y) * c + y
        pts.append((nx, ny))
    return pts


def _cross(ax, ay, bx, by):
    return ax * by - ay * bx


def _intersection(p1, p0, q1, q0):
    """segment p0-p1 with q0-q1 (OpenPCDet `intersection`); returns point or None"""
    # fast rectangle exclusion
    if not (min(p0[0], p1[0]) <= max(q0[0], q1[0]) and min(q0[0], q1[0]) <= max(p0[0], p1[0]) and
            min(p0[1], p1[1]) <= max(q0[1], q1[1]) and min(q0[1], q1[1]) <= max(p0[1], p1[1])):
        return None
    s1 = _cross(q0[0] - p0[0], q0[1] - p0[1], p1[0] - p0[0], p1[1] - p0[1])
    s2 = _cross(p1[0] - p0[0], p1[1] - p0[1], q1[0] - p0[0], q1[1] - p0[1])
    s3 = _cross(p0[0] - q0[0], p0[1] - q0[1], q1[0] - q0[0], q1[1] - q0[1])
    s4 = _cross(q1[0] - q0[0], q1[1] - q0[1], p1[0] - q0[0], p1[1] - q0[1])
    if not (s1 * s2 > 0 and s3 * s4 > 0):
        return None
    s5 = _cross(q1[0] - p0[0], q1[1] - p0[1], p1[0] - p0[0], p1[1] - p0[1])
    if abs(s5 - s1) > EPS:
        return ((s5 * q0[0] - s1 * q1[0]) / (s5 - s1), (s5 * q0[1] - s1 * q1[1]) / (s5 - s1))
    a0, a1 = p0[1] - p1[1], q0[1] - q1[1]
    b0, b1 = p1[0] - p0[0], q1[0] - q0[0]
    c0, c1 = p0[0] * p1[1] - p1[0] * p0[1], q0[0] * q1[1] - q1[0] * q0[1]
    D = a0 * b1 - a1 * b0
    return ((b0 * c1 - b1 * c0) / D, (a1 * c0 - a0 * c1) / D)


def _in_box(box, p):
    cx, cy, dx, dy, ang = float(box[0]), float(box[1]), float(box[3]), float(box[4]), float(box[6])
    c, s = math.cos(-ang), math.sin(-ang)
    rx = (p[0] - cx) * c + (p[1] - cy) * (-s)
    ry = (p[0] - cx) * s + (p[1] - cy) * c
    return abs(rx) < dx / 2 + MARGIN and abs(ry) < dy / 2 + MARGIN


def box_overlap_bev(a, b):
    ca, cb = _corners(a), _corners(b)
    ca.append(ca[0]); cb.append(cb[0])
    pts = []
    for i in range(4):
        for j in range(4):
            p = _intersection(ca[i + 1], ca[i], cb[j + 1], cb[j])
            if p is not None:
                pts.append(p)
    for k in range(4):
        if _in_box(a, cb[k]):
            pts.append(cb[k])
        if _in_box(b, ca[k]):
            pts.append(ca[k])
    if len(pts) < 3:
        return 0.0
    cx = sum(p[0] for p in pts) / len(pts)
    cy = sum(p[1] for p in pts) / len(pts)
    pts.sort(key=lambda p: math.atan2(p[1] - cy, p[0] - cx))
    area = 0.0
    for k in range(len(pts) - 1):
        area += _cross(pts[k][0] - pts[0][0], pts[k][1] - pts[0][1], pts[k + 1][0] - pts[0][0], pts[k + 1][1] - pts[0][1])
    return abs(area) / 2.0


def iou_bev(a, b):
    sa, sb = float(a[3]) * float(a[4]), float(b[3]) * float(b[4])
    so = box_overlap_bev(a, b)
    return so / max(sa + sb - so, EPS)


def nms3d(boxes, scores, thr):
    """greedy NMS, returns kept indices (into boxes) in descending score order (ties: lower index first)"""
    order = torch.argsort(scores, descending=True, stable=True).tolist()
    b = boxes.numpy()
    keep = []
    for i in order:
        if all(iou_bev(b[j], b[i]) <= thr for j in keep):
            keep.append(i)
    return torch.tensor(keep, dtype=torch.long)


def predict_single(level_preds, nms_pre=1000, score_thr=0.01, iou_thr=0.5):
    """fcaf3d_head.py:1352-1399 for one sample.  level_preds: list over levels of (center (n,1), bbox (n,12),
    cls (n,C), points (n,3)).  Returns boxes (M,9), scores (M,), labels (M,)."""
    mb, ms = [], []
    for center, bbox, cls, point in level_preds:
        scores = cls.sigmoid() * center.sigmoid()
        max_scores, _ = scores.max(dim=1)
        if len(scores) > nms_pre > 0:
            ids = torch.argsort(max_scores, descending=True, stable=True)[:nms_pre]
            ids = torch.sort(ids).values                 # keep row order (a set; the reference's topk order is free)
            bbox, scores, point = bbox[ids], scores[ids], point[ids]
        mb.append(G.bbox_pred_to_bbox(point, bbox))
        ms.append(scores)
    boxes, scores = torch.cat(mb), torch.cat(ms)
    out_b, out_s, out_l = [], [], []
    for c in range(scores.shape[1]):
        ids = torch.nonzero(scores[:, c] > score_thr).squeeze(1)
        if ids.numel() == 0:
            continue
        keep = nms3d(boxes[ids][:, :7], scores[ids, c], iou_thr)
        kept = boxes[ids][keep].clone()
        kept[:, 7:] = 0        # fcaf3d_head.py:1681-1682 + euler_box3d.py:44-48: beta, gamma do not survive the NMS wrapper
        out_b.append(kept); out_s.append(scores[ids, c][keep])
        out_l.append(torch.full((len(keep),), c, dtype=torch.long))
    if out_b:
        return torch.cat(out_b), torch.cat(out_s), torch.cat(out_l)
    return boxes.new_zeros((0, 9)), boxes.new_zeros((0,)), torch.zeros((0,), dtype=torch.long)
