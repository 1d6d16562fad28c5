"""Sparse-tensor operators on the CPU (PyTorch f32, differentiable).  TEST ORACLE.

Restates the MinkowskiEngine v0.5.4 operator semantics the reference relies on
(ME is an un-vendored dependency, so this is a
spec-by-restatement, "parity unpinned"): MinkowskiConvolution (k=3 / k=1, stride
1/2), MinkowskiGenerativeConvolutionTranspose (k=2,s=2), MinkowskiMaxPooling
(k=2,s=2), MinkowskiInstanceNorm, MinkowskiBatchNorm, sparse `+`, ME.cat,
MinkowskiPruning, features_at_coordinates.
"""
from dataclasses import dataclass
import numpy as np
import torch
import torch.nn.functional as F
from . import coords as C


@dataclass
class SpT:
    coords: np.ndarray      # (N,4) int32, batch-major
    feats: torch.Tensor     # (N,C) f32
    ts: int                 # tensor stride in original-grid units
    n_batch: int
    cache: dict             # shared coordinate-map cache (the "coordinate manager")

    def new(self, feats, coords=None, ts=None):
        return SpT(self.coords if coords is None else coords, feats,
                   self.ts if ts is None else ts, self.n_batch, self.cache)

    def batch_rows(self, b):
        return np.nonzero(self.coords[:, 0] == b)[0]


def _cached(x, key, fn):
    k = (id(x.coords),) + key
    if k not in x.cache:
        x.cache[k] = fn()
        x.cache.setdefault('_keep', []).append(x.coords)   # keep id() stable
    return x.cache[k]


def gather_conv(feats, nbr, weight):
    """out[j] = sum_k feats[nbr[j,k]] @ weight[k]  (nbr == -1 contributes 0); operands rounded in oracle.rounding's bf16 mode"""
    from . import rounding as R
    return R.op(lambda f, w: _gather_conv(f, nbr, w), feats, weight, feats.shape[1], weight.shape[-1])


def _gather_conv(feats, nbr, weight):
    n_out, K = nbr.shape
    out = feats.new_zeros((n_out, weight.shape[-1]))
    nbr_t = torch.from_numpy(nbr.astype(np.int64))
    for k in range(K):
        col = nbr_t[:, k]
        rows = torch.nonzero(col >= 0).squeeze(1)
        if rows.numel() == 0:
            continue
        out = out.index_add(0, rows, feats[col[rows]] @ weight[k])
    return out


def conv(x, weight, ksize, stride=1, bias=None):
    """MinkowskiConvolution.  weight (K^3, Cin, Cout) or (Cin, Cout) for ksize 1."""
    w = weight if weight.dim() == 3 else weight[None]
    if stride == 1:
        out_coords, out_ts = x.coords, x.ts
    else:
        out_ts = x.ts * stride
        out_coords = _cached(x, ('stride', out_ts), lambda: C.stride_coords(x.coords, out_ts))
    if ksize == 1 and stride == 1:
        from . import rounding as R
        out = R.op(lambda f, ww: f @ ww, x.feats, w[0], x.feats.shape[1], w.shape[-1])
    else:
        nbr = _cached(x, ('kmap', ksize, stride), lambda: C.kernel_map(x.coords, out_coords, ksize, x.ts))
        out = gather_conv(x.feats, nbr, w)
    if bias is not None:
        out = out + bias
    return x.new(out, out_coords, out_ts)


def gen_conv_transpose(x, weight):
    """MinkowskiGenerativeConvolutionTranspose(k=2, s=2); weight (8, Cin, Cout);
    child row 8*i+k = x[i] @ weight[k]."""
    out_coords = _cached(x, ('gen',), lambda: C.gen_transpose_coords(x.coords, x.ts))
    from . import rounding as R
    out = R.op(lambda f, w: torch.einsum('nc,kcd->nkd', f, w).reshape(-1, w.shape[-1]), x.feats, weight, x.feats.shape[1],
               weight.shape[-1])
    return x.new(out, out_coords, x.ts // 2)


def max_pool(x):
    """MinkowskiMaxPooling(k=2, s=2): max over existing voxels of the window
    out + {0,1}^3*ts; first tap wins ties (gradient goes to that one)."""
    out_ts = x.ts * 2
    out_coords = _cached(x, ('stride', out_ts), lambda: C.stride_coords(x.coords, out_ts))
    nbr = _cached(x, ('pmap',), lambda: C.kernel_map(x.coords, out_coords, 2, x.ts))
    nbr_t = torch.from_numpy(nbr.astype(np.int64))
    g = x.feats[nbr_t.clamp(min=0)]                          # (N_out, 8, C)
    g = torch.where((nbr_t >= 0)[..., None], g, torch.full_like(g, float('-inf')))
    return x.new(g.max(dim=1).values, out_coords, out_ts)


def instance_norm(x, weight, bias, eps=1e-8):
    """MinkowskiInstanceNorm: per-sample, per-channel, biased variance,
    1/sqrt(var + 1e-8) (ME MinkowskiInstanceNormFunction), affine (1,C)."""
    out = torch.empty_like(x.feats)
    outs = []
    for b in range(x.n_batch):
        rows = torch.from_numpy(x.batch_rows(b))
        f = x.feats[rows]
        mean = f.mean(0, keepdim=True)
        var = ((f - mean) ** 2).mean(0, keepdim=True)
        outs.append((rows, (f - mean) / torch.sqrt(var + eps)))
    out = x.feats.new_zeros(x.feats.shape)
    for rows, v in outs:
        out = out.index_copy(0, rows, v)
    return x.new(out * weight.view(1, -1) + bias.view(1, -1))


def batch_norm(x, weight, bias, running_mean, running_var, training=True, momentum=0.1, eps=1e-5):
    """MinkowskiBatchNorm == nn.BatchNorm1d over the (N,C) feature matrix."""
    return x.new(F.batch_norm(x.feats, running_mean, running_var, weight, bias, training, momentum, eps))


def union_add(a, b):
    """sparse a + b on the coordinate union (fcaf3d_head.py:1009)."""
    assert a.ts == b.ts
    coords, pa, pb = C.union_coords(a.coords, b.coords, a.n_batch)
    out = a.feats.new_zeros((coords.shape[0], a.feats.shape[1]))
    out = out.index_add(0, torch.from_numpy(pa), a.feats)
    out = out.index_add(0, torch.from_numpy(pb), b.feats)
    return a.new(out, coords, a.ts)


def prune(x, mask):
    keep = np.nonzero(mask)[0]
    return x.new(x.feats[torch.from_numpy(keep)], np.ascontiguousarray(x.coords[keep]), x.ts)


def features_at_coordinates(table, query_coords):
    idx, w = C.interp_weights(query_coords, table.coords, table.ts)
    idx_t = torch.from_numpy(idx.astype(np.int64))
    g = table.feats[idx_t.clamp(min=0)]                      # (N,8,C)
    wt = torch.from_numpy(w) * (idx_t >= 0).float()
    out = table.feats.new_zeros((idx.shape[0], table.feats.shape[1]))
    for k in range(8):                                       # fixed summation order
        out = out + g[:, k] * wt[:, k:k + 1]
    return out
