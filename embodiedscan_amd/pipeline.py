"""Device-side data path for one scan (rows A1-A3): depth maps + camera matrices + the PointSample /
augmentation decisions -> the (n_points,3) augmented global point cloud the detector consumes.
Follows the train pipeline of configs/detection/mv-det3d_...py:134-160; the per-view 4x4 inverses
(torch.inverse of the padded intrinsics, points.py / utils.py:357-359; the camera->global solve of
multiview.py:151-153 as an explicit inverse) are prepared on the host like the reference does."""
import numpy as np
import torch
from .hip import P, call
from .structures import Det3DDataSample, EulerDepthInstance3DBoxes, InstanceData


def scan_matrices(scan):
    V = scan['intrinsic'].shape[0]
    mats = torch.empty((V, 32), dtype=torch.float32)
    for v in range(V):
        pad = torch.eye(4)
        k = torch.from_numpy(scan['intrinsic'][v])
        pad[:k.shape[0], :k.shape[1]] = k
        mats[v, :16] = torch.inverse(pad).reshape(-1)
        mats[v, 16:] = torch.inverse(torch.from_numpy(scan['extrinsic'][v])).reshape(-1)
    a = scan['aug']
    aug = torch.zeros(15, dtype=torch.float32)
    aug[:9] = torch.from_numpy(np.asarray(a['rot'], np.float32)).reshape(-1)
    aug[9] = float(a['scale'])
    aug[10:13] = torch.from_numpy(np.asarray(a['trans'], np.float32))
    aug[13], aug[14] = float(a['hflip']), float(a['vflip'])
    return mats, aug


def upload_scan(scan, device):
    """host -> HBM copy of the raw inputs of one scan (done before the timed region in bench.py)."""
    mats, aug = scan_matrices(scan)
    return dict(depth=torch.from_numpy(scan['depth']).to(device), img=torch.from_numpy(scan['img']).to(device),
                sel_view=torch.from_numpy(scan['sel_view']).to(device), sel_pix=torch.from_numpy(scan['sel_pix']).to(device),
                mats=mats.to(device), aug=aug.to(device), meta=scan['meta'],
                gt_boxes=torch.from_numpy(scan['gt_boxes']), gt_labels=torch.from_numpy(scan['gt_labels']))


_DEV_KEYS = ('depth', 'img', 'sel_view', 'sel_pix', 'mats', 'aug')


def pin_scan(scan):
    """raw inputs of one scan as PINNED host tensors (what a data-loader worker hands over): source of the per-step
    host->device copy that bench.py keeps inside the timed step"""
    mats, aug = scan_matrices(scan)
    host = dict(depth=torch.from_numpy(scan['depth']), img=torch.from_numpy(scan['img']),
                sel_view=torch.from_numpy(scan['sel_view']), sel_pix=torch.from_numpy(scan['sel_pix']), mats=mats, aug=aug)
    out = {k: v.contiguous().pin_memory() for k, v in host.items()}
    out.update(meta=scan['meta'], gt_boxes=torch.from_numpy(scan['gt_boxes']), gt_labels=torch.from_numpy(scan['gt_labels']))
    return out


def alloc_slot(pinned, device):
    """preallocated device buffers shaped like one pinned scan (double-buffered by the caller: no allocation and no
    allocator traffic on the copy stream)"""
    return {k: torch.empty_like(pinned[k], device=device) for k in _DEV_KEYS}


def upload_into(slot, pinned):
    """async host->device copy of one scan into a slot on the CURRENT stream; returns the dscan dict make_batch takes"""
    for k in _DEV_KEYS:
        slot[k].copy_(pinned[k], non_blocking=True)
    return dict(slot, meta=pinned['meta'], gt_boxes=pinned['gt_boxes'], gt_labels=pinned['gt_labels'])


def scan_h2d_bytes(pinned):
    return sum(pinned[k].numel() * pinned[k].element_size() for k in _DEV_KEYS)


def depth_to_points(dscan):
    depth = dscan['depth']
    V, H, W = depth.shape
    n = dscan['sel_pix'].numel()
    out = torch.empty((n, 3), dtype=torch.float32, device=depth.device)
    call('es_depth_to_points', P(depth), H, W, P(dscan['sel_view']), P(dscan['sel_pix']), n, P(dscan['mats']),
         P(dscan['aug']), P(out), torch.cuda.current_stream().cuda_stream)
    return out


def augment_gt_boxes(boxes, aug):
    """Ground-truth side of RandomFlip3D + GlobalRotScaleTrans (augmentation.py:140-168,322-420) for (G,9) Euler boxes,
    host-side like the reference (a few dozen boxes per scan).  The POINT side of the same augmentation runs inside
    es_depth_to_points.  Follows the reference's box class literally -- flips edit the Euler angles in place
    (alpha -> pi - alpha, gamma -> -gamma for X; alpha -> -alpha, beta -> pi - beta for Y: euler_box3d.py:263-281),
    which for a tilted box is not its exact mirror image; the rotation composes matrices and re-extracts ZXY angles.
    aug: dict(hflip, vflip, rot = rot_mat_T as stored in `pcd_rotation`, scale, trans)."""
    import math
    from .geometry import euler_to_matrix_zxy, matrix_to_euler_zxy
    b = torch.as_tensor(boxes, dtype=torch.float32).clone()
    if b.shape[0] == 0:
        return b
    sx, sy = (-1.0 if aug['hflip'] else 1.0), (-1.0 if aug['vflip'] else 1.0)
    xyz = b[:, :3] * torch.tensor([sx, sy, 1.0])
    alpha, beta, gamma = b[:, 6], b[:, 7], b[:, 8]
    if aug['hflip']:
        alpha, gamma = math.pi - alpha, -gamma
    if aug['vflip']:
        alpha, beta = -alpha, math.pi - beta
    rot = torch.as_tensor(aug['rot'], dtype=torch.float32)                  # R^T
    ang = matrix_to_euler_zxy(torch.matmul(rot.t()[None], euler_to_matrix_zxy(torch.stack([alpha, beta, gamma], 1))))
    s, t = float(aug['scale']), torch.as_tensor(aug['trans'], dtype=torch.float32)
    return torch.cat([(xyz @ rot) * s + t, b[:, 3:6] * s, ang], 1)


def make_batch(dscans):
    """-> the `data` dict of mmengine's train_step: {'inputs': {'points', 'img'}, 'data_samples'}."""
    points = [depth_to_points(d) for d in dscans]
    imgs = torch.stack([d['img'] for d in dscans])
    samples = [Det3DDataSample(d['meta'], InstanceData(bboxes_3d=EulerDepthInstance3DBoxes(d['gt_boxes']),
                                                       labels_3d=d['gt_labels'])) for d in dscans]
    return {'inputs': {'points': points, 'img': imgs}, 'data_samples': samples}


def make_occ_batch(dscans, occ_gts):
    """`data` dict for DenseFusionOccPredictor.train_step: the detection batch plus `gt_occupancy` (N,4) and
    `gt_occupancy_masks` (X,Y,Z) on every data sample (Pack3DDetInputs, datasets/transforms/formatting.py:254-264)."""
    data = make_batch(dscans)
    for ds, occ in zip(data['data_samples'], occ_gts):
        ds.gt_occupancy = torch.as_tensor(occ['gt_occupancy'])
        m = occ.get('gt_occupancy_masks')
        ds.gt_occupancy_masks = None if m is None else torch.as_tensor(m)
    return data


def make_grounding_batch(dscans, anns):
    """`data` dict for SparseFeatureFusion3DGrounder.train_step: the detection batch with the prompt (`text`), the
    positive character spans (`tokens_positive`) and the TARGET boxes of the prompt as gt_instances_3d."""
    data = make_batch(dscans)
    for ds, a in zip(data['data_samples'], anns):
        ds.text, ds.tokens_positive = a['text'], a['tokens_positive']
        ds.gt_instances_3d = InstanceData(bboxes_3d=EulerDepthInstance3DBoxes(torch.as_tensor(a['gt_boxes'])),
                                          labels_3d=torch.as_tensor(a['gt_labels']))
    return data
