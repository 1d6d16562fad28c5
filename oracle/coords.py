"""Integer coordinate work of the sparse path, on the CPU (numpy).  TEST ORACLE.

Restates the MinkowskiEngine (v0.5.4, un-vendored dependency of the reference)
coordinate-manager semantics used by the reference at:
  * embodiedscan/models/detectors/sparse_featfusion_single_stage.py:109-118
    (p / voxel_size -> ME.utils.batch_sparse_collate -> ME.SparseTensor)
  * embodiedscan/models/backbones/mink_resnet.py:58-74,122-140 (strided conv /
    pooling coordinate maps)
  * embodiedscan/models/dense_heads/fcaf3d_head.py:937-947,1006-1010,1091-1114
    (generative transpose conv, union add, pruning)

Conventions fixed here (the spec both this oracle and the HIP kernels implement):
  * a coordinate row is int32 (b, x, y, z) in units of the ORIGINAL voxel grid;
  * rows of every coordinate set are kept batch-major; the root voxel set is ordered along a Z-curve
    (Morton key of x,y,z), every derived set in "first occurrence" order of the operation that created it;
  * tap enumeration for a k^3 kernel: x fastest, i.e. k = ix + K*iy + K*K*iz with
    offset (ix,iy,iz) - (K//2 if K odd else 0).
"""
import numpy as np

OFF = 1 << 17      # coordinate bias so that packed fields are non-negative
FIELD = 18         # bits per spatial field


def pack(c):
    """(N,4) int -> (N,) int64 key; lexicographic in (b,x,y,z)."""
    c = np.asarray(c).astype(np.int64)
    assert c.ndim == 2 and c.shape[1] == 4
    if c.shape[0]:
        assert (np.abs(c[:, 1:]) < OFF).all(), 'coordinate out of packable range'
    return (c[:, 0] << (3 * FIELD)) | ((c[:, 1] + OFF) << (2 * FIELD)) | \
        ((c[:, 2] + OFF) << FIELD) | (c[:, 3] + OFF)


def unpack(k):
    k = np.asarray(k).astype(np.int64)
    m = (1 << FIELD) - 1
    return np.stack([k >> (3 * FIELD), ((k >> (2 * FIELD)) & m) - OFF,
                     ((k >> FIELD) & m) - OFF, (k & m) - OFF], 1).astype(np.int32)


def unique_first(keys):
    """Unique keys in first-occurrence order.

    Returns (ukeys, first_index, inverse) with ukeys[inverse] == keys."""
    keys = np.asarray(keys)
    if keys.shape[0] == 0:
        z = np.zeros((0,), np.int64)
        return keys.copy(), z, z
    uk, first, inv = np.unique(keys, return_index=True, return_inverse=True)
    order = np.argsort(first, kind='stable')
    rank = np.empty_like(order)
    rank[order] = np.arange(order.shape[0])
    return uk[order], first[order].astype(np.int64), rank[inv].astype(np.int64)


def lookup(table_keys, query_keys):
    """Row index of each query key inside table_keys (unique), or -1."""
    table_keys = np.asarray(table_keys)
    query_keys = np.asarray(query_keys)
    out = np.full(query_keys.shape, -1, np.int32)
    if table_keys.shape[0] == 0 or query_keys.shape[0] == 0:
        return out
    order = np.argsort(table_keys, kind='stable')
    sk = table_keys[order]
    pos = np.searchsorted(sk, query_keys)
    pos_c = np.minimum(pos, sk.shape[0] - 1)
    hit = sk[pos_c] == query_keys
    out[hit] = order[pos_c[hit]].astype(np.int32)
    return out


def _spread3(v):
    v = v.astype(np.uint64) & np.uint64(0x1fffff)
    for sh, m in ((32, 0x1f00000000ffff), (16, 0x1f0000ff0000ff), (8, 0x100f00f00f00f00f), (4, 0x10c30c30c30c30c3),
                  (2, 0x1249249249249249)):
        v = (v | (v << np.uint64(sh))) & np.uint64(m)
    return v


def morton_key(c):
    """Z-curve key of (N,4) int coords: batch major, then bit-interleaved (x,y,z) (x most significant)."""
    c = np.asarray(c).astype(np.int64)
    m = _spread3(c[:, 3] + OFF) | (_spread3(c[:, 2] + OFF) << np.uint64(1)) | (_spread3(c[:, 1] + OFF) << np.uint64(2))
    return (c[:, 0].astype(np.uint64) << np.uint64(54)) | m


def voxelize(points_list, voxel_size):
    """A4.  sparse_featfusion_single_stage.py:109-118 + ME sparse_collate.

    c = int32(trunc(p / voxel_size)) with a TRUE f32 division, batch
    index prepended, duplicates removed keeping the FIRST point of each voxel
    (SURVEY Q2).  Returns coords (M,4) int32 and src (M,) int64 row index into
    the concatenated point list."""
    cs = []
    for b, p in enumerate(points_list):
        p = np.asarray(p, np.float32)
        q = (p[:, :3] / np.float32(voxel_size)).astype(np.float32)
        c = np.trunc(q).astype(np.int32)
        cs.append(np.concatenate([np.full((c.shape[0], 1), b, np.int32), c], 1))
    c = np.concatenate(cs, 0) if cs else np.zeros((0, 4), np.int32)
    uk, first, _ = unique_first(pack(c))
    # rows of the root voxel set are laid out along a Z-curve (a free choice: ME's order is hash-map order); every
    # derived set keeps "first occurrence" order and therefore inherits the spatial locality
    order = np.argsort(morton_key(c[first]), kind='stable')
    first = first[order]
    return c[first], first


def kernel_offsets(ksize, scale):
    """(K^3, 3) int offsets in original-grid units, x fastest."""
    r = np.arange(ksize) - (ksize // 2 if ksize % 2 == 1 else 0)
    iz, iy, ix = np.meshgrid(r, r, r, indexing='ij')
    return (np.stack([ix.ravel(), iy.ravel(), iz.ravel()], 1) * scale).astype(np.int32)


def stride_coords(coords, out_ts):
    """Output coordinates of a strided conv / pool: floor(c / out_ts) * out_ts,
    unique, first-occurrence order (true floor for negatives; ME semantics)."""
    c = np.asarray(coords, np.int32).copy()
    c[:, 1:] = np.floor_divide(c[:, 1:], out_ts) * out_ts
    uk, first, _ = unique_first(pack(c))
    return c[first]


def kernel_map(in_coords, out_coords, ksize, in_ts):
    """A6.  nbr[j, k] = row of in_coords at out_coords[j] + offset_k * in_ts, or -1."""
    offs = kernel_offsets(ksize, in_ts)
    tk = pack(in_coords)
    nbr = np.full((out_coords.shape[0], offs.shape[0]), -1, np.int32)
    for k, o in enumerate(offs):
        q = np.asarray(out_coords, np.int32).copy()
        q[:, 1:] += o[None]
        nbr[:, k] = lookup(tk, pack(q))
    return nbr


def inverse_map(nbr, n_in):
    """inv[i, k] = output row j with nbr[j, k] == i, or -1 (unique by construction)."""
    inv = np.full((n_in, nbr.shape[1]), -1, np.int32)
    j, k = np.nonzero(nbr >= 0)
    inv[nbr[j, k], k] = j.astype(np.int32)
    return inv


def gen_transpose_coords(coords, in_ts):
    """MinkowskiGenerativeConvolutionTranspose(k=2, s=2): every voxel at tensor
    stride in_ts emits 8 children c + {0,1}^3 * in_ts/2; child row = 8*i + k."""
    half = in_ts // 2
    offs = kernel_offsets(2, half)
    c = np.repeat(np.asarray(coords, np.int32), 8, axis=0)
    c[:, 1:] += np.tile(offs, (coords.shape[0], 1))
    return c


def batch_counts(coords, n_batch):
    return np.bincount(coords[:, 0], minlength=n_batch).astype(np.int64)


def union_coords(a, b, n_batch):
    """Coordinate union for sparse `a + b` (fcaf3d_head.py:1009).

    Result is batch-major; inside a batch: all rows of `a`, then the rows of `b`
    that are not in `a`, each in original order.  Returns (coords, pos_a, pos_b)
    where pos_x[i] is the union row of x's row i."""
    ka, kb = pack(a), pack(b)
    hit = lookup(ka, kb)                      # row of a for each b row
    new = hit < 0
    ca = batch_counts(a, n_batch)
    cn = np.bincount(b[new, 0], minlength=n_batch).astype(np.int64)
    tot = ca + cn
    base = np.concatenate([[0], np.cumsum(tot)[:-1]])
    a_start = np.concatenate([[0], np.cumsum(ca)[:-1]])
    pos_a = base[a[:, 0]] + (np.arange(a.shape[0]) - a_start[a[:, 0]])
    pos_b = np.empty(b.shape[0], np.int64)
    # rank of each new b row within its batch
    nb = b[new, 0]
    n_start = np.concatenate([[0], np.cumsum(cn)[:-1]])
    rank = np.arange(nb.shape[0]) - n_start[nb]
    pos_b[new] = base[nb] + ca[nb] + rank
    pos_b[~new] = pos_a[hit[~new]]
    out = np.empty((int(tot.sum()), 4), np.int32)
    out[pos_a] = a
    out[pos_b[new]] = b[new]
    return out, pos_a.astype(np.int64), pos_b.astype(np.int64)


def interp_weights(query_coords, table_coords, table_ts):
    """features_at_coordinates / MinkowskiInterpolation (fcaf3d_head.py:1102-1103).

    For each query (integer original-grid coordinate, used as float) the 8 corners
    lower + {0,1}^3*ts with lower = floor(q/ts)*ts and trilinear weights
    prod(1 - |q - corner| / ts); absent corners are index -1.
    Returns idx (N,8) int32, w (N,8) float32."""
    q = np.asarray(query_coords, np.int32)
    lower = q.copy()
    lower[:, 1:] = np.floor_divide(q[:, 1:], table_ts) * table_ts
    frac = ((q[:, 1:] - lower[:, 1:]).astype(np.float32) / np.float32(table_ts))
    offs = kernel_offsets(2, table_ts)
    tk = pack(table_coords)
    idx = np.full((q.shape[0], 8), -1, np.int32)
    w = np.zeros((q.shape[0], 8), np.float32)
    for k, o in enumerate(offs):
        c = lower.copy()
        c[:, 1:] += o[None]
        idx[:, k] = lookup(tk, pack(c))
        sel = (o > 0)
        wk = np.ones(q.shape[0], np.float32)
        for d in range(3):
            wk = wk * (frac[:, d] if sel[d] else (np.float32(1) - frac[:, d]))
        w[:, k] = wk
    return idx, w
