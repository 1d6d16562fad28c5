"""CPU restatement of the DEVICE-side PointSample draws (embodiedscan_amd/csrc/data.hip: es_draw_keys / es_draw_keys_index, used by
ScanPipeline(device_draws=True)).  TEST ORACLE (imported by tests/ only).

The reference draws `np.random.choice(range(n), k, replace=False)` per depth frame and once more over the aggregated cloud
(datasets/transforms/points.py:155-213, multiview.py:139-169, configs/detection/mv-det3d_...py:141-143).  Its law: a uniformly
random k-subset in uniformly random order.  The device path realises the same law with a counter-based generator instead of
numpy's sequential stream (whose Fisher-Yates over all ~3e5 valid pixels of every frame was 52 % of the host time per scan,
profiles/r3_loader_profile.txt): element i of stream s gets the 30-bit key u = top 30 bits of splitmix64((seed ^ s * C1) + i * C2);
the k LARGEST keys are selected (ties: lower index -- a tie AT the threshold has probability ~ n / 2^30 = 3e-4 per frame and then
prefers the earlier pixel: a bias far below anything a training run can see) and emitted in descending key order (ties: lower
index).  Selecting the k largest of n i.i.d. uniform keys is a uniform k-subset; sorting them by key is a uniform random order
(tests/test_draws.py checks inclusion and position frequencies).  Every function here mirrors one kernel, integer for integer."""
import numpy as np

C1 = np.uint64(0x9E3779B97F4A7C15)
C2 = np.uint64(0xD1B54A32D192ED03)
M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(x):
    x = x.astype(np.uint64)
    with np.errstate(over='ignore'):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & M64
        x = ((x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & M64
        x = ((x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & M64
        return x ^ (x >> np.uint64(31))


def key30(seed, stream, index):
    """30-bit key of element `index` (array) of stream `stream` under `seed` (es_draw_key in csrc/data.hip)"""
    with np.errstate(over='ignore'):
        s = (np.uint64(seed) ^ (np.uint64(stream) * C1)) & M64
        x = (s + index.astype(np.uint64) * C2) & M64
    return (splitmix64(x) >> np.uint64(34)).astype(np.int64)


def draw(keys, valid, k):
    """indices of the k valid elements with the largest keys, in (descending key, ascending index) order; needs >= k valid"""
    idx = np.flatnonzero(valid)
    assert len(idx) >= k
    order = np.lexsort((idx, -keys[idx]))[:k]                    # primary: key descending, secondary: index
    return idx[order]


def point_sample(depth, seed, view_points, n_points):
    """depth (V, H, W) -> (sel_view, sel_pix) int32 of length n_points: per view `view_points` of the non-zero pixels, then
    `n_points` of the V * view_points aggregated ones (streams 0..V-1 for the views, stream 255 for the aggregate).
    Requires every view to hold >= view_points valid pixels and V * view_points >= n_points (the caller falls back to the host
    draws otherwise: `replace=True` cases of the reference)."""
    V = depth.shape[0]
    d = depth.reshape(V, -1)
    pix = np.arange(d.shape[1], dtype=np.int64)
    sv, sp = [], []
    for v in range(V):
        sel = draw(key30(seed, v, pix), d[v] != 0, view_points)
        sv.append(np.full(view_points, v, np.int32))
        sp.append(sel.astype(np.int32))
    sv, sp = np.concatenate(sv), np.concatenate(sp)
    j = np.arange(len(sp), dtype=np.int64)
    pick = draw(key30(seed, 255, j), np.ones(len(sp), bool), n_points)
    return sv[pick], sp[pick]
