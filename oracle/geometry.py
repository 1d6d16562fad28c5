"""Rotation / box geometry + target assignment + box losses on the CPU.  TEST ORACLE.

Plain PyTorch f32 restatement of (reference file:line):
  * pytorch3d.transforms.euler_angles_to_matrix / matrix_to_euler_angles, convention
    'ZXY' (un-vendored dependency; used at structures/bbox_3d/utils.py:67,
    losses/chamfer_distance.py:179-181, dense_heads/fcaf3d_head.py:1516)
  * rotation_3d_in_euler            structures/bbox_3d/utils.py:32-86
  * ortho_6d_2_Mat                  dense_heads/fcaf3d_head.py:1728-1750
  * _bbox_pred_to_bbox (12-d)       dense_heads/fcaf3d_head.py:1454-1525
  * _get_face_distances / _get_centerness / get_targets
                                    dense_heads/fcaf3d_head.py:1527-1664
  * bbox_to_corners / chamfer_distance / BBoxCDLoss
                                    losses/chamfer_distance.py:13-79,160-285
  * EulerInstance3DBoxes.corners    structures/bbox_3d/euler_box3d.py:142-184
"""
import torch


def euler_to_matrix_zxy(a):
    """R = Rz(a0) @ Rx(a1) @ Ry(a2); a (...,3) -> (...,3,3)."""
    ca, sa = torch.cos(a[..., 0]), torch.sin(a[..., 0])
    cb, sb = torch.cos(a[..., 1]), torch.sin(a[..., 1])
    cc, sc = torch.cos(a[..., 2]), torch.sin(a[..., 2])
    one, zero = torch.ones_like(ca), torch.zeros_like(ca)
    rz = torch.stack([ca, -sa, zero, sa, ca, zero, zero, zero, one], -1).reshape(a.shape[:-1] + (3, 3))
    rx = torch.stack([one, zero, zero, zero, cb, -sb, zero, sb, cb], -1).reshape(a.shape[:-1] + (3, 3))
    ry = torch.stack([cc, zero, sc, zero, one, zero, -sc, zero, cc], -1).reshape(a.shape[:-1] + (3, 3))
    return torch.matmul(torch.matmul(rz, rx), ry)


def matrix_to_euler_zxy(m):
    """Inverse of euler_to_matrix_zxy (pytorch3d Tait-Bryan branch for 'ZXY')."""
    a0 = torch.atan2(-m[..., 0, 1], m[..., 1, 1])
    a1 = torch.asin(m[..., 2, 1])
    a2 = torch.atan2(-m[..., 2, 0], m[..., 2, 2])
    return torch.stack([a0, a1, a2], -1)


def rotation_3d_in_euler(points, angles):
    """points (N,M,3), angles (N,3) -> points @ R(angles)^T   (utils.py:67-76)."""
    rot_t = euler_to_matrix_zxy(angles).transpose(-2, -1)
    if points.shape[0] == 0:
        return points
    return torch.bmm(points, rot_t)


def ortho_6d_2_mat(x_raw, y_raw):
    """fcaf3d_head.py:1728-1750 (Gram-Schmidt, eps 1e-8 added to the norm)."""
    def normalize(v):
        return v / (torch.norm(v, dim=1, keepdim=True) + 1e-8)
    y = normalize(y_raw)
    z = normalize(torch.cross(x_raw, y, dim=1))
    x = torch.cross(y, z, dim=1)
    return torch.cat((x.unsqueeze(2), y.unsqueeze(2), z.unsqueeze(2)), 2)


def bbox_pred_to_bbox(points, bbox_pred):
    """A13.  12-d prediction -> (N,9) box.  fcaf3d_head.py:1454-1525."""
    if bbox_pred.shape[0] == 0:
        return bbox_pred
    shift = torch.stack(((bbox_pred[:, 1] - bbox_pred[:, 0]) / 2,
                         (bbox_pred[:, 3] - bbox_pred[:, 2]) / 2,
                         (bbox_pred[:, 5] - bbox_pred[:, 4]) / 2), dim=-1).view(-1, 1, 3)
    rot_mat = ortho_6d_2_mat(bbox_pred[:, 6:9], bbox_pred[:, 9:])
    euler = matrix_to_euler_zxy(rot_mat)
    shift = rotation_3d_in_euler(shift, euler)[:, 0, :]
    center = points + shift
    size = torch.stack((bbox_pred[:, 0] + bbox_pred[:, 1], bbox_pred[:, 2] + bbox_pred[:, 3],
                        bbox_pred[:, 4] + bbox_pred[:, 5]), dim=-1)
    return torch.cat((center, size, euler), dim=-1)


def bbox_to_corners(bbox):
    """chamfer_distance.py:160-203; 9-DoF branch."""
    rot = euler_to_matrix_zxy(bbox[:, 6:9])
    sx = bbox.new_tensor([1, 1, 1, 1, -1, -1, -1, -1])
    sy = bbox.new_tensor([1, 1, -1, -1, 1, 1, -1, -1])
    sz = bbox.new_tensor([1, -1, 1, -1, 1, -1, 1, -1])
    signs = torch.stack((sx, sy, sz), -1)                    # (8,3)
    corners = signs[None] * (bbox[:, None, 3:6] / 2)         # (N,8,3)
    return bbox[:, None, :3] + torch.matmul(corners, rot.transpose(1, 2))


def bbox_cd_loss(source, target, loss_weight=1.0):
    """A15.  BBoxCDLoss(mode='l1', group='g8', reduction='mean')
    chamfer_distance.py:265-285 + :13-79: L1 corner distance matrix (N,8,8), min over
    target corners, mean over N*8."""
    sc, tc = bbox_to_corners(source), bbox_to_corners(target)
    dist = (sc[:, :, None, :] - tc[:, None, :, :]).abs().sum(-1)
    return dist.min(dim=2).values.mean() * loss_weight


def euler_box_corners(boxes):
    """EulerInstance3DBoxes.corners, euler_box3d.py:142-184."""
    import numpy as np
    cn = torch.from_numpy(np.stack(np.unravel_index(np.arange(8), [2] * 3), axis=1)).to(boxes.dtype)
    cn = cn[[0, 1, 3, 2, 4, 5, 7, 6]] - 0.5
    corners = boxes[:, 3:6].view(-1, 1, 3) * cn.reshape(1, 8, 3)
    corners = rotation_3d_in_euler(corners, boxes[:, 6:9])
    return corners + boxes[:, :3].view(-1, 1, 3)


def _rot_rows(angles):
    """The 9 entries of R(angles) (ZXY), each (G,), computed with explicit
    elementwise f32 ops in a fixed order (shared spec with the HIP kernel, which
    receives these matrices precomputed on the host)."""
    r = euler_to_matrix_zxy(angles)
    return r


def face_distances(points, boxes):
    """fcaf3d_head.py:1527-1557.  points (N,3), boxes (G,9) -> (N,G,6).

    shift' = shift @ R(-euler)^T is evaluated as an explicit left-to-right
    mul/add chain (no FMA, no BLAS) so that it is bit-reproducible on the GPU:
        s'_c = s_0*R[c,0] + s_1*R[c,1] + s_2*R[c,2]
    The reference uses torch.bmm here, whose summation order is backend-defined."""
    rot = euler_to_matrix_zxy(-boxes[:, 6:9])                # (G,3,3)  
    s = points[:, None, :] - boxes[None, :, :3]              # (N,G,3)
    sh = []
    for c in range(3):
        sh.append((s[..., 0] * rot[None, :, c, 0] + s[..., 1] * rot[None, :, c, 1])
                  + s[..., 2] * rot[None, :, c, 2])
    cen = [boxes[None, :, d] + sh[d] for d in range(3)]
    out = []
    for d in range(3):
        half = boxes[None, :, 3 + d] / 2
        out.append(cen[d] - boxes[None, :, d] + half)
        out.append(boxes[None, :, d] + half - cen[d])
    return torch.stack(out, -1)


def centerness_from_faces(fd):
    """fcaf3d_head.py:1559-1576 (left-to-right product, then sqrt)."""
    x, y, z = fd[..., 0:2], fd[..., 2:4], fd[..., 4:6]
    c = x.min(-1)[0] / x.max(-1)[0] * y.min(-1)[0] / y.max(-1)[0] * z.min(-1)[0] / z.max(-1)[0]
    # IEEE correctly-rounded f32 sqrt (sqrt in f64, then one rounding).  torch's CPU sqrt on large f32 tensors
    # goes through MKL VML and is off by 1 ulp in ~0.6% of the elements (and exact on short tensors), so it
    # cannot serve as a reproducible spec; the HIP kernel uses the correctly rounded sqrtf.
    if c.dtype == torch.float32:
        return torch.sqrt(c.double()).float()
    return torch.sqrt(c)


@torch.no_grad()
def get_targets(points_per_level, gt_boxes, gt_labels, pts_assign_threshold=27,
                pts_center_threshold=18):
    """A12.  fcaf3d_head.py:1578-1664 for EulerDepthInstance3DBoxes (with_yaw=True,
    gravity_center == tensor[:, :3], volume = w*l*h: euler_box3d.py:137-140,
    base_box3d.py:87-90).  Returns center_targets (N,), bbox_targets (N,9),
    cls_targets (N,) int64 in [-1, n_cls)."""
    float_max = 1e8
    n_levels = len(points_per_level)
    levels = torch.cat([torch.full((len(p),), i, dtype=torch.int64) for i, p in enumerate(points_per_level)])
    points = torch.cat(points_per_level)
    n_points, n_boxes = len(points), len(gt_boxes)
    if n_boxes == 0:
        return (points.new_zeros((n_points,)), points.new_zeros((n_points, 9)),
                gt_labels.new_full((n_points,), -1))
    volumes = (gt_boxes[:, 3] * gt_boxes[:, 4] * gt_boxes[:, 5])[None].expand(n_points, n_boxes)
    fd = face_distances(points, gt_boxes)
    inside = fd.min(dim=-1).values > 0
    n_pos = torch.stack([inside[levels == i].sum(0) for i in range(n_levels)], 0)
    lower_limit_mask = n_pos < pts_assign_threshold
    lower_index = torch.argmax(lower_limit_mask.int(), dim=0) - 1
    lower_index = torch.where(lower_index < 0, 0, lower_index)
    all_upper = torch.all(~lower_limit_mask, dim=0)
    best_level = torch.where(all_upper, n_levels - 1, lower_index)
    level_cond = best_level[None, :] == levels[:, None]
    cen = centerness_from_faces(fd)
    cen = torch.where(inside, cen, torch.full_like(cen, -1))
    cen = torch.where(level_cond, cen, torch.full_like(cen, -1))
    top = torch.topk(cen, min(pts_center_threshold + 1, len(cen)), dim=0).values[-1]
    topk_cond = cen > top[None]
    vol = torch.where(inside, volumes, torch.full_like(volumes, float_max))
    vol = torch.where(level_cond, vol, torch.full_like(vol, float_max))
    vol = torch.where(topk_cond, vol, torch.full_like(vol, float_max))
    min_vol, min_inds = vol.min(dim=1)
    ar = torch.arange(n_points)
    center_targets = cen[ar, min_inds]
    bbox_targets = gt_boxes[min_inds]
    cls_targets = gt_labels[min_inds]
    cls_targets = torch.where(min_vol == float_max, torch.full_like(cls_targets, -1), cls_targets)
    return center_targets, bbox_targets, cls_targets


def sigmoid_focal_loss_sum(logits, labels, gamma=2.0, alpha=0.25):
    """mmcv sigmoid_focal_loss semantics (un-vendored): label outside
    [0,C) means an all-negative row.  Returns the SUM over all elements."""
    n, c = logits.shape
    t = torch.zeros_like(logits)
    valid = (labels >= 0) & (labels < c)
    t[torch.nonzero(valid).squeeze(1), labels[valid]] = 1
    p = torch.sigmoid(logits)
    pt = (1 - p) * t + p * (1 - t)
    w = (alpha * t + (1 - alpha) * (1 - t)) * pt.pow(gamma)
    bce = torch.nn.functional.binary_cross_entropy_with_logits(logits, t, reduction='none')
    return (bce * w).sum()
