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


def _axis_table(n_src, n_dst):
    """first source index + the two 11-bit weights per output index of OpenCV's INTER_LINEAR (imgproc resize.cpp,
    resizeGeneric_: fx = (float)((d + 0.5) * scale - 0.5), clamped at both ends, cvRound(w * 2048))"""
    f = ((np.arange(n_dst, dtype=np.float64) + 0.5) * (float(n_src) / float(n_dst)) - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int32)
    f = (f - s.astype(np.float32)).astype(np.float32)
    edge = (s < 0) | (s >= n_src - 1)
    s = np.clip(s, 0, n_src - 1)
    f[edge] = 0.0
    c = np.stack([np.float32(1.0) - f, f], 1) * np.float32(2048.0)
    return torch.from_numpy(s), torch.from_numpy(np.rint(c).astype(np.int16))


_RESIZE_TABLES = {}


def resize_tables(H, W, h, w, device):
    """device-resident coefficient tables of es_resize_u8 for (H, W) -> (h, w), built once per size pair and device"""
    key = (H, W, h, w, str(device))
    t = _RESIZE_TABLES.get(key)
    if t is None:
        xo, xa = _axis_table(W, w)
        yo, yb = _axis_table(H, h)
        t = _RESIZE_TABLES[key] = tuple(a.to(device) for a in (xo, xa, yo, yb))
    return t


def resize_frames(img_raw, size_hw, out=None):
    """Resize(keep_ratio=False) of the decoded frames on the device: (V,H,W,3) u8 RGB -> (V,3,h,w) u8 (es_resize_u8)"""
    V, H, W, _ = img_raw.shape
    h, w = size_hw
    if out is None:
        out = torch.empty((V, 3, h, w), dtype=torch.uint8, device=img_raw.device)
    xo, xa, yo, yb = resize_tables(H, W, h, w, img_raw.device)
    call('es_resize_u8', P(img_raw), V, H, W, P(xo), P(xa), P(yo), P(yb), h, w, P(out),
         torch.cuda.current_stream().cuda_stream)
    return out


def _host_tensors(scan):
    """the arrays of one raw scan that go to the device.  A scan from the dataset reader carries `img_raw` (decoded frames
    at the file resolution, resized on the device); a synthetic scan carries `img` at the network resolution."""
    mats, aug = scan_matrices(scan)
    # (the two big arrays first: a loader worker decodes straight into its slot and must know their offsets before the small
    #  arrays -- whose presence depends on the decoded depth -- exist; datasets/loader.py `alloc`)
    host = dict(depth=torch.from_numpy(scan['depth']))
    if 'img_raw' in scan:
        host['img_raw'] = torch.from_numpy(scan['img_raw'])
    else:
        host['img'] = torch.from_numpy(scan['img'])
    host.update(mats=mats, aug=aug)
    if 'sel_pix' in scan:                    # (absent: the PointSample draws are made on the device, see device_point_sample)
        host.update(sel_view=torch.from_numpy(scan['sel_view']), sel_pix=torch.from_numpy(scan['sel_pix']))
    return host


def _finish(d, scan):
    d.update(meta=scan['meta'], gt_boxes=torch.as_tensor(scan['gt_boxes']), gt_labels=torch.as_tensor(scan['gt_labels']))
    for k in ('gt_occupancy', 'gt_occupancy_masks', 'visible_occupancy_masks', 'visible_instance_masks', 'point_range',
              'text', 'tokens_positive', 'draw'):
        if k in scan:
            d[k] = scan[k]
    return d


def upload_scan(scan, device):
    """host -> HBM copy of the raw inputs of one scan (blocking; bench.py's timed path uses pin_scan / upload_into)."""
    d = {k: v.to(device) for k, v in _host_tensors(scan).items()}
    if 'img_raw' in d:
        d['img'] = resize_frames(d.pop('img_raw'), scan['meta']['img_shape'])
    return _ensure_draws(_finish(d, scan))


def pin_scan(scan, pin=True):
    """raw inputs of one scan as PINNED host tensors (what a data-loader worker hands over): source of the per-step
    host->device copy that bench.py keeps inside the timed step"""
    out = {k: (v.contiguous().pin_memory() if pin else v.contiguous()) for k, v in _host_tensors(scan).items()}
    return _finish(out, scan)


def _dev_keys(pinned):
    return [k for k in ('depth', 'img', 'img_raw', 'sel_view', 'sel_pix', 'mats', 'aug') if k in pinned]


def alloc_slot(pinned, device):
    """preallocated device buffers shaped like one pinned scan (double-buffered by the caller: no allocation and no
    allocator traffic on the copy stream)"""
    slot = {k: torch.empty_like(pinned[k], device=device) for k in _dev_keys(pinned)}
    if 'img_raw' in slot:
        h, w = pinned['meta']['img_shape']
        slot['img'] = torch.empty((pinned['img_raw'].shape[0], 3, h, w), dtype=torch.uint8, device=device)
    return slot


def upload_into(slot, pinned):
    """async host->device copy of one scan into a slot on the CURRENT stream (+ the device resize of decoded frames, on
    the same stream); returns the dscan dict make_batch takes"""
    for k in _dev_keys(pinned):
        slot[k].copy_(pinned[k], non_blocking=True)
    if 'img_raw' in slot:
        resize_frames(slot['img_raw'], pinned['meta']['img_shape'], out=slot['img'])
    d = dict(slot)
    d.pop('img_raw', None)
    return _ensure_draws(_finish(d, {k: pinned[k] for k in pinned if k not in slot}))


# ------------------------------------------------------------------ N4: PointSample on the device
_DRAW_WS = {}


def device_point_sample(depth, seed, view_points, n_points):
    """PointSample(num_points=view_points) per depth frame + PointSample(n_points) over the aggregated cloud
    (datasets/transforms/points.py:155-213, configs/detection/mv-det3d_...py:141-143) ON THE DEVICE, from counter-based keys
    (csrc/data.hip; the law of the reference's draws, not its numpy stream; oracle/draws.py restates it integer for integer).
    depth (V, H, W) f32 on the device; every view must hold >= view_points non-zero pixels and V * view_points >= n_points
    (the host checks both and falls back to host draws otherwise).  -> (sel_view, sel_pix) int32 (n_points,), random order."""
    from . import hip
    V, H, W = depth.shape
    HW, M = H * W, V * view_points
    dev = depth.device
    st = torch.cuda.current_stream().cuda_stream
    i32, i64 = torch.int32, torch.int64

    def topk(values, seg, k, mask):
        ns = len(seg) - 1
        key = (st, 'topk')
        ws = _DRAW_WS.get(key)
        need = int(hip.raw('es_topk_mask_workspace_ints')(32))
        if ws is None or ws.device != dev:
            ws = _DRAW_WS[key] = torch.zeros(need, dtype=i32, device=dev)
        call('es_topk_mask_ws', values, hip.iarr(seg), ns, k, mask, P(ws), ws.numel(), st)

    def select_sorted(values, keys, n, seg_fn, k_total):
        """keys of the selected elements (mask from per-segment top-k) in ascending key order -> (k_total,) int64"""
        mask = torch.empty(n, dtype=i32, device=dev)
        seg_fn(mask)
        scratch = torch.empty(n + n // 2048 + 8, dtype=i32, device=dev)
        sel = torch.empty(k_total, dtype=i64, device=dev)
        src = torch.empty(k_total, dtype=i32, device=dev)
        call('es_compact_mask', P(keys), n, P(mask), P(scratch), P(sel), P(src), 0, st)
        nb = int(hip.raw('es_sort_scratch_bytes')(k_total))
        sc = torch.empty(nb, dtype=torch.uint8, device=dev)
        out = torch.empty(k_total, dtype=i64, device=dev)
        call('es_sort_u64', P(sel), 0, k_total, P(sc), nb, P(out), 0, st)
        return out

    values = torch.empty(V * HW, dtype=torch.float32, device=dev)
    keys = torch.empty(V * HW, dtype=i64, device=dev)
    call('es_draw_keys', P(depth), V, HW, int(seed), P(values), P(keys), st)

    def per_view(mask):
        for v0 in range(0, V, 32):                         # es_topk_mask_ws: at most ES_MAX_SEG = 32 segments per call
            nv = min(32, V - v0)
            topk(values.data_ptr() + 4 * v0 * HW, [i * HW for i in range(nv + 1)], view_points, mask.data_ptr() + 4 * v0 * HW)
    k1 = select_sorted(values, keys, V * HW, per_view, M)
    view1, pix1 = torch.empty(M, dtype=i32, device=dev), torch.empty(M, dtype=i32, device=dev)
    call('es_draw_unpack', P(k1), M, 0, 0, P(view1), P(pix1), st)
    v2, kk2 = torch.empty(M, dtype=torch.float32, device=dev), torch.empty(M, dtype=i64, device=dev)
    call('es_draw_keys_index', M, int(seed), 255, P(v2), P(kk2), st)
    k2 = select_sorted(v2, kk2, M, lambda mask: topk(P(v2), [0, M], n_points, P(mask)), n_points)
    sel_view, sel_pix = torch.empty(n_points, dtype=i32, device=dev), torch.empty(n_points, dtype=i32, device=dev)
    call('es_draw_unpack', P(k2), n_points, P(view1), P(pix1), P(sel_view), P(sel_pix), st)
    return sel_view, sel_pix


def _ensure_draws(d):
    """a scan handed over WITHOUT sel_view / sel_pix (ScanPipeline(device_draws=True)) gets them drawn here, on the stream of the upload"""
    if 'sel_pix' not in d and d.get('draw') is not None:
        seed, vp, npts = d['draw']
        d['sel_view'], d['sel_pix'] = device_point_sample(d['depth'], seed, vp, npts)
    return d


def scan_h2d_bytes(pinned):
    return sum(pinned[k].numel() * pinned[k].element_size() for k in _dev_keys(pinned))


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
    st = dscans[0].get('_img_stack') if dscans else None       # upload_batch: the frames already are one (B, V, 3, h, w) block
    imgs = st if (st is not None and st.shape[0] == len(dscans)) else torch.stack([d['img'] for d in dscans])
    samples = [Det3DDataSample(d['meta'], InstanceData(bboxes_3d=EulerDepthInstance3DBoxes(d['gt_boxes']),
                                                       labels_3d=d['gt_labels'])) for d in dscans]
    return {'inputs': {'points': points, 'img': imgs}, 'data_samples': samples}


def make_occ_batch(dscans, occ_gts=None):
    """`data` dict for DenseFusionOccPredictor.train_step: the detection batch plus `gt_occupancy` (N,4) and
    `gt_occupancy_masks` (X,Y,Z) on every data sample (Pack3DDetInputs, datasets/transforms/formatting.py:254-264).
    occ_gts None: scans from the dataset reader carry both themselves."""
    data = make_batch(dscans)
    if occ_gts is None:
        occ_gts = [dict(gt_occupancy=d['gt_occupancy'], gt_occupancy_masks=d.get('gt_occupancy_masks')) for d in dscans]
    for ds, occ in zip(data['data_samples'], occ_gts):
        ds.gt_occupancy = torch.as_tensor(occ['gt_occupancy'])
        m = occ.get('gt_occupancy_masks')
        ds.gt_occupancy_masks = None if m is None else torch.as_tensor(m)
    return data


def make_grounding_batch(dscans, anns=None):
    """`data` dict for SparseFeatureFusion3DGrounder.train_step: the detection batch with the prompt (`text`), the
    positive character spans (`tokens_positive`) and the TARGET boxes of the prompt as gt_instances_3d.
    anns None: scans from MultiView3DGroundingDataset carry text / spans / target boxes themselves."""
    data = make_batch(dscans)
    if anns is None:
        anns = [dict(text=d['text'], tokens_positive=d['tokens_positive'], gt_boxes=d['gt_boxes'], gt_labels=d['gt_labels'])
                for d in dscans]
    for ds, a in zip(data['data_samples'], anns):
        ds.text, ds.tokens_positive = a['text'], a['tokens_positive']
        ds.gt_instances_3d = InstanceData(bboxes_3d=EulerDepthInstance3DBoxes(torch.as_tensor(a['gt_boxes'])),
                                          labels_3d=torch.as_tensor(a['gt_labels']))
    return data


# ------------------------------------------------------------------ one slab per batch (bench.py, loader hand-over)
# A batch of B scans used to go up as 7 tensors per scan (28+ hipMemcpyAsync per step, most of them a few hundred bytes).
# Here every array of every scan of the batch lives in ONE pinned byte slab (256-byte aligned pieces, arrays of one kind
# adjacent so that equal-shaped frames are also one (B, V, 3, h, w) view) and the device side is one byte slab of the same
# layout: the host->device copy of a batch is a single hipMemcpyAsync.
_SLAB_ALIGN = 256


class BatchSlab:
    """views[i][key] -> tensor of scan i inside `slab` (uint8, pinned host or device memory); `layout` = [(scan, key,
    offset, shape, dtype)]; `extras[i]` = the host-side fields of scan i (meta, ground truth, prompt ...)."""

    def __init__(self, slab, layout, extras):
        self.slab, self.layout, self.extras = slab, layout, extras
        self.views = [dict() for _ in extras]
        for i, k, off, shape, dtype in layout:
            n = 1
            for s in shape:
                n *= int(s)
            nb = n * torch.empty((), dtype=dtype).element_size()
            self.views[i][k] = slab[off:off + nb].view(dtype).view(shape)

    @property
    def nbytes(self):
        return int(self.slab.numel())

    def stacked(self, key):
        """(B, ...) view over the `key` arrays of all scans when they are equal-shaped and adjacent, else None"""
        ent = [e for e in self.layout if e[1] == key]
        if not ent or any(e[3] != ent[0][3] for e in ent):
            return None
        n = 1
        for s in ent[0][3]:
            n *= int(s)
        nb = n * torch.empty((), dtype=ent[0][4]).element_size()
        stride = ent[1][2] - ent[0][2] if len(ent) > 1 else nb
        if stride != nb or any(b[2] - a[2] != stride for a, b in zip(ent[:-1], ent[1:])):
            return None
        return self.slab[ent[0][2]:ent[0][2] + nb * len(ent)].view(ent[0][4]).view((len(ent),) + tuple(ent[0][3]))


def pin_batch(scans, pin=True):
    """the raw inputs of a batch of scans in ONE pinned host slab (-> BatchSlab)"""
    hosts = [_host_tensors(s) for s in scans]
    keys = [k for k in ('img', 'img_raw', 'depth', 'sel_view', 'sel_pix', 'mats', 'aug') if any(k in h for h in hosts)]
    layout, off = [], 0
    for k in keys:                       # arrays of one kind adjacent; a kind starts on an aligned offset
        off = -(-off // _SLAB_ALIGN) * _SLAB_ALIGN
        for i, h in enumerate(hosts):
            if k not in h:
                continue
            t = h[k]
            layout.append((i, k, off, tuple(t.shape), t.dtype))
            nb = t.numel() * t.element_size()
            # equal-shaped arrays stay adjacent (stackable) when their size keeps the alignment, else pad each piece
            off += nb if nb % 16 == 0 else -(-nb // _SLAB_ALIGN) * _SLAB_ALIGN
    total = -(-off // _SLAB_ALIGN) * _SLAB_ALIGN
    slab = torch.empty(max(total, _SLAB_ALIGN), dtype=torch.uint8)
    if pin:
        slab = slab.pin_memory()
    extras = [_finish({}, s) for s in scans]
    out = BatchSlab(slab, layout, extras)
    for i, h in enumerate(hosts):
        for k, t in h.items():
            out.views[i][k].copy_(t)
    return out


def alloc_batch_slot(nbytes, device):
    """device byte slab able to hold any batch of up to `nbytes` (+ the resized frames of file-backed scans are allocated
    per upload by resize_frames)"""
    return torch.empty(int(nbytes), dtype=torch.uint8, device=device)


def upload_batch(slot, pinned):
    """ONE async host->device copy of a whole batch on the current stream -> list of dscan dicts (make_batch's input);
    equal-shaped frames additionally come back as one stacked (B, V, 3, h, w) tensor under '_img_stack' of scan 0"""
    assert slot.numel() >= pinned.nbytes, 'device slot smaller than the batch slab'
    dst = slot[:pinned.nbytes]
    dst.copy_(pinned.slab, non_blocking=True)
    dev = BatchSlab(dst, pinned.layout, pinned.extras)
    out = []
    for i, v in enumerate(dev.views):
        d = dict(v)
        if 'img_raw' in d:
            d['img'] = resize_frames(d.pop('img_raw'), pinned.extras[i]['meta']['img_shape'])
        d.update(pinned.extras[i])
        out.append(_ensure_draws(d))
    st = dev.stacked('img')
    if st is not None and out:
        out[0]['_img_stack'] = st
    return out
