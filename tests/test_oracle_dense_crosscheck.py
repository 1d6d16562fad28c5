"""The part of the oracle that restates MinkowskiEngine (an absent third-party dependency, "parity unpinned") is anchored
here on an independent implementation of the published operator semantics: PyTorch's DENSE conv3d / conv_transpose3d /
max_pool3d / batch_norm on the voxelised grid.  A sparse convolution is by definition the dense convolution restricted to
the occupied output sites with absent inputs contributing zero, so on a small grid the two must agree to f32 rounding.
The kernel-offset order (x fastest) and the (K, C_in, C_out) weight layout are OUR stated convention; the test maps them
onto conv3d's (C_out, C_in, kD=z, kH=y, kW=x) explicitly."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import sparse as S

G = 12        # dense grid edge (voxels at tensor stride 1)


def _random_set(seed, n_batch=2, fill=0.25, ts=1, lo=0):
    rng = np.random.default_rng(seed)
    cs = []
    for b in range(n_batch):
        occ = rng.random((G, G, G)) < fill
        x, y, z = np.nonzero(occ)
        c = np.stack([np.full_like(x, b), x + lo, y + lo, z + lo], 1).astype(np.int32)
        c[:, 1:] *= ts
        cs.append(c[rng.permutation(len(c))])
    return np.concatenate(cs)


def _to_dense(coords, feats, ts, n_batch, lo=0, fill=0.0):
    """(B, C, Z, Y, X) dense tensor (conv3d layout D=z, H=y, W=x)"""
    Cn = feats.shape[1]
    d = torch.full((n_batch, Cn, G, G, G), fill, dtype=feats.dtype)
    c = torch.from_numpy(coords.astype(np.int64))
    xi, yi, zi = c[:, 1] // ts - lo, c[:, 2] // ts - lo, c[:, 3] // ts - lo
    d[c[:, 0], :, zi, yi, xi] = feats
    return d


def _gather_dense(d, coords, ts, lo=0):
    c = torch.from_numpy(coords.astype(np.int64))
    return d[c[:, 0], :, c[:, 3] // ts - lo, c[:, 2] // ts - lo, c[:, 1] // ts - lo]


def _w_dense(w, k):
    """(K^3, Cin, Cout) with x-fastest taps -> conv3d weight (Cout, Cin, kz, ky, kx)"""
    return w.reshape(k, k, k, w.shape[1], w.shape[2]).permute(4, 3, 0, 1, 2).contiguous()


@pytest.mark.parametrize('lo', [0, -5])          # -5: negative coordinates on every axis
def test_conv3_stride1_equals_dense_conv3d(lo):
    coords = _random_set(1, lo=lo)
    g = torch.Generator().manual_seed(2)
    x = torch.randn(len(coords), 5, generator=g, dtype=torch.float64)
    w = torch.randn(27, 5, 7, generator=g, dtype=torch.float64)
    y = S.conv(S.SpT(coords, x, 1, 2, {}), w, 3, 1).feats
    dense = F.conv3d(_to_dense(coords, x, 1, 2, lo), _w_dense(w, 3), padding=1)
    ref = _gather_dense(dense, coords, 1, lo)
    assert torch.allclose(y, ref, rtol=1e-12, atol=1e-12)


def test_conv3_stride2_equals_dense_conv3d():
    """ME: output sites = floor(c / 2) * 2 of the occupied inputs; output j sums inputs at j + {-1,0,1}^3 (input stride)"""
    coords = _random_set(3, fill=0.15)
    g = torch.Generator().manual_seed(4)
    x = torch.randn(len(coords), 4, generator=g, dtype=torch.float64)
    w = torch.randn(27, 4, 6, generator=g, dtype=torch.float64)
    out = S.conv(S.SpT(coords, x, 1, 2, {}), w, 3, 2)
    assert out.ts == 2
    exp_sites = np.unique(np.concatenate([coords[:, :1], coords[:, 1:] // 2 * 2], 1), axis=0)
    assert sorted(map(tuple, out.coords)) == sorted(map(tuple, exp_sites))
    dense = F.conv3d(_to_dense(coords, x, 1, 2), _w_dense(w, 3), padding=1)      # value at every stride-1 site
    ref = _gather_dense(dense, out.coords, 1)                                      # read at the even sites
    assert torch.allclose(out.feats, ref, rtol=1e-12, atol=1e-12)


def test_generative_transpose_equals_dense_conv_transpose3d():
    coords = _random_set(5, fill=0.1, ts=2)[:, :]
    coords = coords[(coords[:, 1:] < G).all(1)]                # children must stay inside the dense grid
    g = torch.Generator().manual_seed(6)
    x = torch.randn(len(coords), 3, generator=g, dtype=torch.float64)
    w = torch.randn(8, 3, 5, generator=g, dtype=torch.float64)
    out = S.gen_conv_transpose(S.SpT(coords, x, 2, 2, {}), w)
    assert out.ts == 1 and len(out.coords) == 8 * len(coords)
    xd = torch.zeros((2, 3, G // 2, G // 2, G // 2), dtype=torch.float64)
    c = torch.from_numpy(coords.astype(np.int64))
    xd[c[:, 0], :, c[:, 3] // 2, c[:, 2] // 2, c[:, 1] // 2] = x
    wd = w.reshape(2, 2, 2, 3, 5).permute(3, 4, 0, 1, 2).contiguous()            # (Cin, Cout, kz, ky, kx)
    dense = F.conv_transpose3d(xd, wd, stride=2)
    ref = _gather_dense(dense, out.coords, 1)
    assert torch.allclose(out.feats, ref, rtol=1e-12, atol=1e-12)


def test_max_pool_equals_dense_max_pool3d():
    coords = _random_set(7, fill=0.3)
    g = torch.Generator().manual_seed(8)
    x = torch.randn(len(coords), 6, generator=g, dtype=torch.float64)
    out = S.max_pool(S.SpT(coords, x, 1, 2, {}))
    dense = F.max_pool3d(_to_dense(coords, x, 1, 2, fill=float('-inf')), 2, 2)   # absent voxels never win
    c = torch.from_numpy(out.coords.astype(np.int64))
    ref = dense[c[:, 0], :, c[:, 3] // 2, c[:, 2] // 2, c[:, 1] // 2]
    assert torch.equal(out.feats, ref)


def test_norms_equal_torch_functional():
    coords = _random_set(9)
    g = torch.Generator().manual_seed(10)
    x = torch.randn(len(coords), 8, generator=g, dtype=torch.float64) * 3 + 1
    wt, b = torch.rand(8, generator=g, dtype=torch.float64) + 0.5, torch.randn(8, generator=g, dtype=torch.float64)
    st = S.SpT(coords, x, 1, 2, {})
    y = S.instance_norm(st, wt, b).feats
    for bi in range(2):
        rows = torch.from_numpy(st.batch_rows(bi))
        ref = F.instance_norm(x[rows].t()[None], weight=wt, bias=b, eps=1e-8)[0].t()
        assert torch.allclose(y[rows], ref, rtol=1e-10, atol=1e-10)
    rm, rv = torch.zeros(8, dtype=torch.float64), torch.ones(8, dtype=torch.float64)
    y = S.batch_norm(st, wt, b, rm.clone(), rv.clone()).feats
    mean, var = x.mean(0), x.var(0, unbiased=False)
    assert torch.allclose(y, (x - mean) / torch.sqrt(var + 1e-5) * wt + b, rtol=1e-10, atol=1e-10)


def test_union_add_equals_dense_add():
    a, bc = _random_set(11), _random_set(12)
    g = torch.Generator().manual_seed(13)
    fa = torch.randn(len(a), 4, generator=g, dtype=torch.float64)
    fb = torch.randn(len(bc), 4, generator=g, dtype=torch.float64)
    cache = {}
    out = S.union_add(S.SpT(a, fa, 1, 2, cache), S.SpT(bc, fb, 1, 2, cache))
    dense = _to_dense(a, fa, 1, 2) + _to_dense(bc, fb, 1, 2)
    assert torch.allclose(out.feats, _gather_dense(dense, out.coords, 1))
    assert len(out.coords) == len(np.unique(np.concatenate([a, bc]), axis=0))


def test_features_at_coordinates_equals_dense_trilinear():
    """MinkowskiInterpolation: trilinear weights over the 8 surrounding sites of the coarse grid, absent sites contribute
    zero and the weights are NOT renormalised == grid_sample(bilinear, zeros padding, align_corners) on the dense grid."""
    ts = 2
    table = _random_set(14, fill=0.3, ts=ts)
    table = table[(table[:, 1:] < G).all(1)]
    g = torch.Generator().manual_seed(15)
    f = torch.randn(len(table), 5, generator=g, dtype=torch.float32)
    query = _random_set(16, fill=0.2)
    query = query[(query[:, 1:] < G - ts).all(1)]
    out = S.features_at_coordinates(S.SpT(table, f, ts, 2, {}), query)
    n = G // ts
    d = torch.zeros((2, 5, n, n, n))
    c = torch.from_numpy(table.astype(np.int64))
    d[c[:, 0], :, c[:, 3] // ts, c[:, 2] // ts, c[:, 1] // ts] = f
    q = torch.from_numpy(query.astype(np.float32))
    for b in range(2):
        qb = q[q[:, 0] == b][:, 1:] / ts                                           # coarse-grid units (x, y, z)
        grid = (qb / (n - 1) * 2 - 1).view(1, -1, 1, 1, 3)                          # grid_sample wants (x, y, z) in [-1, 1]
        ref = F.grid_sample(d[b:b + 1], grid, mode='bilinear', padding_mode='zeros', align_corners=True)
        ref = ref[0, :, :, 0, 0].t()
        got = out[torch.from_numpy(query[:, 0] == b)]
        assert torch.allclose(got, ref, rtol=1e-5, atol=1e-6)
