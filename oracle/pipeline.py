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


def augment_points(points, aug):
    """RandomFlip3D then GlobalRotScaleTrans on a (N,3) tensor (augmentation.py:140-168, 322-420; point side:
    depth_points.py:39-50, base_points.py:168-201,215-231,308-315)."""
    out = points.clone()
    if aug['hflip']:
        out[:, 0] = -out[:, 0]
    if aug['vflip']:
        out[:, 1] = -out[:, 1]
    out = out @ torch.as_tensor(aug['rot'], dtype=out.dtype)
    return out * aug['scale'] + torch.as_tensor(aug['trans'], dtype=out.dtype)


def augment_boxes(boxes, aug):
    """The same augmentation on (G,9) Euler boxes, literally as the reference's box class does it:
    flip X: x -> -x, alpha -> pi - alpha, gamma -> -gamma; flip Y: y -> -y, alpha -> -alpha, beta -> pi - beta
    (euler_box3d.py:263-281 -- NOT the exact mirror image of a tilted box); rotate: centre -> R c, angles ->
    euler(R @ R_box) (euler_box3d.py:187-206,216-244); scale multiplies centre and size (:208-214); translate adds to
    the centre (base_box3d.py:248-263).  aug['rot'] is rot_mat_T = R^T as stored in `pcd_rotation`."""
    from . import geometry as G
    import math
    b = boxes.clone()
    if b.shape[0] == 0:
        return b
    if aug['hflip']:
        b[:, 0] = -b[:, 0]
        b[:, 6] = -b[:, 6] + math.pi
        b[:, 8] = -b[:, 8]
    if aug['vflip']:
        b[:, 1] = -b[:, 1]
        b[:, 6] = -b[:, 6]
        b[:, 7] = -b[:, 7] + math.pi
    R = torch.as_tensor(aug['rot'], dtype=b.dtype).t()
    centre = b[:, :3] @ R.t()
    ang = G.matrix_to_euler_zxy(torch.matmul(R[None], G.euler_to_matrix_zxy(b[:, 6:])))
    b = torch.cat([centre, b[:, 3:6], ang], 1)
    b[:, :6] = b[:, :6] * aug['scale']
    b[:, :3] = b[:, :3] + torch.as_tensor(aug['trans'], dtype=b.dtype)
    return b
