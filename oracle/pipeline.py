"""Data-side rows A1-A3 + augmentation on the CPU, PyTorch f32.  TEST ORACLE.

Follows MultiViewPipeline / ConvertRGBDToPoints / PointSample / AggregateMultiViewPoints /
RandomFlip3D / GlobalRotScaleTrans as configured at
configs/detection/mv-det3d_8xb4_embodiedscan-3d-284class-9dof.py:134-160, with the random
decisions (sample indices, flips, angle, scale, translation) taken from the scan dict."""
import torch
from . import model as M


def scan_to_points(scan):
    """-> (n_points,3) f32 augmented global points.  The two PointSample stages are
    pre-composed into (sel_view, sel_pix) by the generator; un-projecting only the
    selected pixels is identical to un-projecting all and indexing."""
    depth = torch.from_numpy(scan['depth'])
    V = depth.shape[0]
    out = torch.empty((len(scan['sel_pix']), 3), dtype=torch.float32)
    sv, sp = torch.from_numpy(scan['sel_view']).long(), torch.from_numpy(scan['sel_pix']).long()
    for v in range(V):
        pts, _ = M.unproject_depth(depth[v], torch.from_numpy(scan['intrinsic'][v]))
        m = torch.nonzero(sv == v).squeeze(1)
        g = M.aggregate_points(pts[sp[m]], torch.from_numpy(scan['extrinsic'][v]))
        out[m] = g
    a = scan['aug']
    if a['hflip']:
        out[:, 0] = -out[:, 0]
    if a['vflip']:
        out[:, 1] = -out[:, 1]
    out = out @ torch.from_numpy(a['rot'])
    out = out * a['scale']
    out = out + torch.from_numpy(a['trans'])
    return out
