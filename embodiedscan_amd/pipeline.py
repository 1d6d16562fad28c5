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


def depth_to_points(dscan):
    depth = dscan['depth']
    V, H, W = depth.shape
    n = dscan['sel_pix'].numel()
    out = torch.empty((n, 3), dtype=torch.float32, device=depth.device)
    call('es_depth_to_points', P(depth), H, W, P(dscan['sel_view']), P(dscan['sel_pix']), n, P(dscan['mats']),
         P(dscan['aug']), P(out), torch.cuda.current_stream().cuda_stream)
    return out


def make_batch(dscans):
    """-> the `data` dict of mmengine's train_step: {'inputs': {'points', 'img'}, 'data_samples'}."""
    points = [depth_to_points(d) for d in dscans]
    imgs = torch.stack([d['img'] for d in dscans])
    samples = [Det3DDataSample(d['meta'], InstanceData(bboxes_3d=EulerDepthInstance3DBoxes(d['gt_boxes']),
                                                       labels_3d=d['gt_labels'])) for d in dscans]
    return {'inputs': {'points': points, 'img': imgs}, 'data_samples': samples}
