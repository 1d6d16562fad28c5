"""Host-side (CPU torch) geometry helpers of the product path: tiny per-box / per-view
matrix preparation that the reference also does on the host before launching kernels
(pytorch3d euler_angles_to_matrix('ZXY') at embodiedscan/structures/bbox_3d/utils.py:67)."""
import torch


def euler_to_matrix_zxy(a):
    """R = Rz(a0) @ Rx(a1) @ Ry(a2); (...,3) -> (...,3,3)."""
    ca, sa = torch.cos(a[..., 0]), torch.sin(a[..., 0])
    cb, sb = torch.cos(a[..., 1]), torch.sin(a[..., 1])
    cc, sc = torch.cos(a[..., 2]), torch.sin(a[..., 2])
    one, zero = torch.ones_like(ca), torch.zeros_like(ca)
    rz = torch.stack([ca, -sa, zero, sa, ca, zero, zero, zero, one], -1).reshape(a.shape[:-1] + (3, 3))
    rx = torch.stack([one, zero, zero, zero, cb, -sb, zero, sb, cb], -1).reshape(a.shape[:-1] + (3, 3))
    ry = torch.stack([cc, zero, sc, zero, one, zero, -sc, zero, cc], -1).reshape(a.shape[:-1] + (3, 3))
    return torch.matmul(torch.matmul(rz, rx), ry)


def matrix_to_euler_zxy(m):
    """inverse of euler_to_matrix_zxy on its principal branch (pytorch3d matrix_to_euler_angles(m, 'ZXY'))"""
    return torch.stack([torch.atan2(-m[..., 0, 1], m[..., 1, 1]), torch.asin(m[..., 2, 1]),
                        torch.atan2(-m[..., 2, 0], m[..., 2, 2])], -1)
