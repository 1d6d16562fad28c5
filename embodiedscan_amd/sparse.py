"""Host side of the coordinate manager: coordinate sets, hash tables and cached kernel maps
living in HBM, built by the es_hip kernels (csrc/coords.hip).

Mirrors what the reference obtains implicitly from MinkowskiEngine's CoordinateManager
(`coordinate_map_key` / `coordinate_manager` at
embodiedscan/models/detectors/sparse_featfusion_single_stage.py:215-218): every sparse tensor
derived from one input shares the maps cached here.
"""
import ctypes
import weakref

import torch
from . import hip
from .hip import P, call


def _stream():
    return hip.stream()


# Row counts come back through PINNED host memory: a device-to-host copy into pageable memory is staged and ordered by the
# runtime in a way that made the read-backs of the next-batch prefetch stream wait for the whole train step queued on the
# other streams (round 4, profiles/r4b_timeline.txt); into pinned memory it is a plain stream-ordered DMA.
_PIN = {}


def _pinned_ints(n=64):
    t = _PIN.get('buf')
    if t is None or t.numel() < n:
        t = torch.empty(max(n, 256), dtype=torch.int32)
        if torch.cuda.is_available():
            t = t.pin_memory()
        _PIN['buf'] = t
    return t


def read_ints(dev_tensor):
    """host list of a small device int32 tensor: async copy into the pinned buffer + a sync of the CURRENT stream only"""
    n = dev_tensor.numel()
    pin = _pinned_ints(n)
    pin[:n].copy_(dev_tensor.reshape(-1), non_blocking=True)
    torch.cuda.current_stream(dev_tensor.device).synchronize()
    return [int(v) for v in pin[:n].tolist()]


def _pow2_cap(n):
    c = 1024
    while c < 2 * n + 2:
        c *= 2
    return c


class CoordSet:
    """A unique, batch-major set of voxel coordinates at tensor stride `ts`."""

    def __init__(self, keys, n, ts, n_batch, tkeys=None, tvals=None):
        self.keys, self.n, self.ts, self.n_batch = keys, int(n), int(ts), int(n_batch)
        self.tkeys, self.tvals = tkeys, tvals
        self._coords = None
        self._off_dev = None
        self._off_host = None
        self.cache = {}

    @property
    def device(self):
        return self.keys.device

    def table(self):
        if self.tkeys is None:
            cap = _pow2_cap(self.n)
            self.tkeys = torch.empty(cap, dtype=torch.int64, device=self.device)
            self.tvals = torch.empty(cap, dtype=torch.int32, device=self.device)
            call('es_build_table', P(self.keys), self.n, P(self.tkeys), P(self.tvals), cap, _stream())
        return self.tkeys, self.tvals, self.tkeys.numel()

    @property
    def coords(self):
        """(n,4) int32 (b,x,y,z) on the device."""
        if self._coords is None:
            self._coords = torch.empty((self.n, 4), dtype=torch.int32, device=self.device)
            call('es_keys_to_coords', P(self.keys), self.n, P(self._coords), _stream())
        return self._coords

    def offsets_dev(self):
        if self._off_dev is None:
            self._off_dev = torch.empty(self.n_batch + 1, dtype=torch.int32, device=self.device)
            call('es_batch_offsets', P(self.keys), self.n, self.n_batch, P(self._off_dev), _stream())
        return self._off_dev

    def offsets(self):
        """host list of n_batch+1 row offsets (one small D2H copy, cached)."""
        if self._off_host is None:
            self._off_host = read_ints(self.offsets_dev())
        return self._off_host

    # ------------------------------------------------------------------ derived sets / maps
    def strided(self, stride):
        key = ('stride', stride)
        if key not in self.cache:
            out_ts = self.ts * stride
            tmp = torch.empty(self.n, dtype=torch.int64, device=self.device)
            call('es_stride_keys', P(self.keys), self.n, out_ts, P(tmp), _stream())
            self.cache[key] = unique_first(tmp, self.n, out_ts, self.n_batch)[0]
        return self.cache[key]

    # Maps are cached per partner set, keyed by id(partner).  The entry holds the partner only through a WEAK reference, checked on
    # every hit (a dead partner's id may be reused by a new set: then the entry is stale and is rebuilt).  A strong reference --
    # rounds 1-4 -- closed reference cycles (a set's own 3x3x3 map refers to the set; parent -> strided child -> map keyed by the
    # parent): every step's coordinate sets, their maps and ~0.8 GB of tensors hanging off them survived until Python's cyclic
    # collector ran, ~every 20 steps, for 110-170 ms of host time (the 3.5x step-time outliers of profiles/r5b_bench_grounding_diag.json,
    # tools/gc_hunt.py), and the allocator kept asking the driver for fresh memory in between.
    def _cached(self, key, partner):
        e = self.cache.get(key)
        return e[0] if (e is not None and e[1]() is partner) else None

    def kernel_map(self, out, ksize):
        """nbr (out.n, ksize^3) int32: rows of self around each row of `out`."""
        key = ('kmap', id(out), ksize)
        nbr = self._cached(key, out)
        if nbr is None:
            tk, tv, cap = self.table()
            K = ksize ** 3
            nbr = torch.empty((out.n, K), dtype=torch.int32, device=self.device)
            call('es_kernel_map', P(out.keys), out.n, P(tk), P(tv), cap, ksize, self.ts, P(nbr), _stream())
            self.cache[key] = (hip.register_map(nbr), weakref.ref(out))
        return nbr

    def inverse_map(self, out, ksize):
        key = ('imap', id(out), ksize)
        inv = self._cached(key, out)
        if inv is None:
            nbr = self.kernel_map(out, ksize)
            K = ksize ** 3
            inv = torch.empty((self.n, K), dtype=torch.int32, device=self.device)
            call('es_inverse_map', P(nbr), out.n, K, self.n, P(inv), _stream())
            if out is self and ksize % 2 == 1:
                # a set's own odd-kernel map: row j is the neighbour of row i under offset d iff i is the neighbour of j under -d, and the
                # taps are ordered so that offset(K - 1 - k) = -offset(k): inv[i][k] == nbr[i][K - 1 - k] (tests/test_gpu_halo.py checks it).
                # The halo kernel runs such a data gradient on the forward map's plan with mirrored taps (engine._halo_launch).
                inv._mirror_of = nbr
            self.cache[key] = (hip.register_map(inv), weakref.ref(out))
        return inv

    def children(self):
        """MinkowskiGenerativeConvolutionTranspose(k=2,s=2) output set: row 8*i+k."""
        if 'gen' not in self.cache:
            keys = torch.empty(self.n * 8, dtype=torch.int64, device=self.device)
            call('es_gen_children_keys', P(self.keys), self.n, self.ts // 2, P(keys), _stream())
            self.cache['gen'] = CoordSet(keys, self.n * 8, self.ts // 2, self.n_batch)
        return self.cache['gen']


COORD_BATCH = [__import__('os').environ.get('ES_COORD_BATCH', '1') != '0']


def strided_chain(root, n_levels):
    """The sets root.strided(2), .strided(2).strided(2), ... (n_levels of them) with their per-sample offsets, in ONE host
    round trip (es_strided_chain: every level straight from the root keys -- same rows, row order and tables as the chain).
    Installs them in the caches the chain would have filled (`a.strided(2)` then finds its child) and returns the list."""
    have, cur = [], root
    for _ in range(n_levels):                               # already built (a second call in the same step)?
        nxt = cur.cache.get(('stride', 2))
        if nxt is None:
            break
        have.append(nxt)
        cur = nxt
    if len(have) == n_levels:
        return have
    dev, n, B = root.device, root.n, root.n_batch
    if not COORD_BATCH[0] or n <= 0 or have:
        out, cur = [], root
        for _ in range(n_levels):
            cur = cur.strided(2)
            out.append(cur)
        return out
    cap = _pow2_cap(n)
    ts = [root.ts * (2 ** (l + 1)) for l in range(n_levels)]
    tk = [torch.empty(cap, dtype=torch.int64, device=dev) for _ in range(n_levels)]
    tv = [torch.empty(cap, dtype=torch.int32, device=dev) for _ in range(n_levels)]
    ok = [torch.empty(n, dtype=torch.int64, device=dev) for _ in range(n_levels)]
    tmp = torch.empty(n, dtype=torch.int64, device=dev)
    scratch = torch.empty(2 * n + n // 2048 + 8, dtype=torch.int32, device=dev)
    per = B + 2
    res = torch.empty(n_levels * per, dtype=torch.int32, device=dev)
    res_host = _pinned_ints(n_levels * per)
    ptrs = lambda ts_: (ctypes.c_void_p * n_levels)(*[t.data_ptr() for t in ts_])
    call('es_strided_chain', P(root.keys), n, B, n_levels, (ctypes.c_int * n_levels)(*ts), P(tmp), P(scratch), ptrs(tk), ptrs(tv),
         (ctypes.c_int * n_levels)(*([cap] * n_levels)), ptrs(ok), P(res), res_host.data_ptr(), _stream())
    res_host = res_host[:n_levels * per].tolist()
    out, cur = [], root
    for l in range(n_levels):
        m = int(res_host[l * per])
        cs = CoordSet(ok[l][:m], m, ts[l], B, tk[l], tv[l])
        cs._off_dev = res[l * per + 1:(l + 1) * per]
        cs._off_host = [int(res_host[l * per + 1 + b]) for b in range(B + 1)]
        cur.cache[('stride', 2)] = cs
        out.append(cs)
        cur = cs
    return out


def unique_first(keys, n, ts, n_batch, want_src=True):
    """hash-unique in first-occurrence order.  Returns (CoordSet, src_rows int32)."""
    dev = keys.device
    cap = _pow2_cap(n)
    tkeys = torch.empty(cap, dtype=torch.int64, device=dev)
    tvals = torch.empty(cap, dtype=torch.int32, device=dev)
    scratch = torch.empty(2 * n + n // 2048 + 8, dtype=torch.int32, device=dev)
    out_keys = torch.empty(max(n, 1), dtype=torch.int64, device=dev)
    out_src = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
    cnt = _pinned_ints()
    call('es_unique_first', P(keys), n, P(tkeys), P(tvals), cap, P(scratch), P(out_keys), P(out_src),
         cnt.data_ptr(), _stream())
    m = int(cnt[0])
    return CoordSet(out_keys[:m], m, ts, n_batch, tkeys, tvals), out_src[:m]


def voxelize(points, voxel_size):
    """A4.  list of (N_i, >=3) f32 device tensors -> (CoordSet at stride 1, src rows into cat(points))."""
    dev = points[0].device
    n = sum(int(p.shape[0]) for p in points)
    keys = torch.empty(n, dtype=torch.int64, device=dev)
    o = 0
    for b, p in enumerate(points):
        assert p.dtype == torch.float32 and p.stride(1) == 1
        call('es_voxel_keys', P(p), p.shape[0], p.stride(0), b, float(voxel_size), keys.data_ptr() + 8 * o, _stream())
        o += int(p.shape[0])
    cs, src = unique_first(keys, n, 1, len(points))
    return morton_sorted(cs, src)


def voxelize_range(points, range_min, voxel_size, clamp_max):
    """occupancy detector voxelisation (dense_fusion_occ.py:227-245): coords = clamp(trunc((p - range_min) / voxel_size),
    0, clamp_max) per axis.  -> (CoordSet at stride 1, src rows into cat(points))."""
    dev = points[0].device
    n = sum(int(p.shape[0]) for p in points)
    keys = torch.empty(n, dtype=torch.int64, device=dev)
    rng = hip.farr(list(range_min) + list(voxel_size) + [float(c) for c in clamp_max])
    o = 0
    for b, p in enumerate(points):
        assert p.dtype == torch.float32 and p.stride(1) == 1
        call('es_voxel_keys_range', P(p), p.shape[0], p.stride(0), b, rng, keys.data_ptr() + 8 * o, _stream())
        o += int(p.shape[0])
    cs, src = unique_first(keys, n, 1, len(points))
    return morton_sorted(cs, src)


def morton_sorted(cs, src):
    """re-order a unique set (and its source rows) along a Z-curve; see csrc/sort.hip."""
    m = cs.n
    if m == 0:
        return cs, src
    dev = cs.device
    nbytes = hip.raw('es_sort_scratch_bytes')(m)
    scratch = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    out_keys = torch.empty(m, dtype=torch.int64, device=dev)
    out_src = torch.empty(m, dtype=torch.int32, device=dev)
    call('es_morton_sort', P(cs.keys), P(src), m, P(scratch), nbytes, P(out_keys), P(out_src), _stream())
    return CoordSet(out_keys, m, cs.ts, cs.n_batch), out_src


def union(a, b):
    """coordinate union for sparse a + b.  Returns (CoordSet, pos_a, pos_b) (int32 rows).  Cached on `a` per partner
    set (like the kernel maps), so a coordinate prefetch pass can pay the row-count read-back early."""
    key = ('union', id(b))
    e = a.cache.get(key)
    if e is None or e[3]() is not b:               # (weak reference to the partner, as for the maps: no reference cycles)
        e = a.cache[key] = _union(a, b) + (weakref.ref(b),)
    return e[:3]


def _union(a, b):
    dev = a.device
    tk, tv, cap = a.table()
    scratch = torch.empty(3 * b.n + b.n // 2048 + 8, dtype=torch.int32, device=dev)
    pos_a = torch.empty(max(a.n, 1), dtype=torch.int32, device=dev)
    pos_b = torch.empty(max(b.n, 1), dtype=torch.int32, device=dev)
    out_keys = torch.empty(a.n + b.n, dtype=torch.int64, device=dev)
    cnt = _pinned_ints()
    call('es_union_plan', P(a.keys), a.n, P(tk), P(tv), cap, P(b.keys), b.n, P(a.offsets_dev()), P(b.offsets_dev()),
         a.n_batch, P(scratch), P(pos_a), P(pos_b), P(out_keys), cnt.data_ptr(), _stream())
    m = int(cnt[0])
    return CoordSet(out_keys[:m], m, a.ts, a.n_batch), pos_a[:a.n], pos_b[:b.n]


def compact(cs, mask, offsets=None):
    """rows of `cs` where mask (int32 0/1) is set.  Returns (CoordSet, src rows).
    offsets: host list of the RESULT's per-sample row offsets when the caller knows them (top-k pruning keeps
    min(n_b, k) rows of sample b): the launch then stays stream-ordered, no row-count read-back."""
    dev = cs.device
    scratch = torch.empty(cs.n + cs.n // 2048 + 8, dtype=torch.int32, device=dev)
    out_keys = torch.empty(max(cs.n, 1), dtype=torch.int64, device=dev)
    out_src = torch.empty(max(cs.n, 1), dtype=torch.int32, device=dev)
    if offsets is not None:
        call('es_compact_mask', P(cs.keys), cs.n, P(mask), P(scratch), P(out_keys), P(out_src), 0, _stream())
        m = int(offsets[-1])
        out = CoordSet(out_keys[:m], m, cs.ts, cs.n_batch)
        out._off_host = [int(v) for v in offsets]
        return out, out_src[:m]
    cnt = _pinned_ints()
    call('es_compact_mask', P(cs.keys), cs.n, P(mask), P(scratch), P(out_keys), P(out_src), cnt.data_ptr(),
         _stream())
    m = int(cnt[0])
    return CoordSet(out_keys[:m], m, cs.ts, cs.n_batch), out_src[:m]


def interp_map(query, table_set):
    tk, tv, cap = table_set.table()
    idx = torch.empty((query.n, 8), dtype=torch.int32, device=query.device)
    w = torch.empty((query.n, 8), dtype=torch.float32, device=query.device)
    call('es_interp_map', P(query.keys), query.n, P(tk), P(tv), cap, table_set.ts, P(idx), P(w), _stream())
    return idx, w


class SparseTensor:
    """Feature matrix `F` (engine.Var, (n,C)) on a CoordSet -- the ME.SparseTensor stand-in."""

    def __init__(self, cs, F):
        self.cs, self.F = cs, F

    @property
    def C(self):                       # ME: x.C -> (n,4) int coordinates
        return self.cs.coords

    @property
    def features(self):
        return self.F.d

    @property
    def tensor_stride(self):
        return self.cs.ts

    @property
    def decomposition_permutations(self):
        off = self.cs.offsets()
        return [torch.arange(off[b], off[b + 1], device=self.cs.device) for b in range(self.cs.n_batch)]

    @property
    def decomposed_coordinates(self):
        off = self.cs.offsets()
        return [self.cs.coords[off[b]:off[b + 1], 1:] for b in range(self.cs.n_batch)]
