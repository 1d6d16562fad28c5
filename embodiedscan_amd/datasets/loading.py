"""Host side of the real-data path: file decode and the random decisions of the reference's train / test
pipeline.  Everything that touches pixels or points afterwards runs on the device:

  reference CPU transform (configs/detection/mv-det3d_...py:134-160)      here
  -----------------------------------------------------------------      ---------------------------------------------
  MultiViewPipeline view choice   (transforms/multiview.py:46-64)         select_views()            host, same draws
  LoadImageFromFile / LoadDepthFromFile (transforms/loading.py:53-81)     decode_image / decode_depth   host (PIL)
  ConvertRGBDToPoints + PointSample(n/10) (transforms/points.py:40-53,    sample_pixels(): only the CHOICE of pixels;
      189-206), AggregateMultiViewPoints, PointSample(n)                  unprojection = es_depth_to_points (A1-A3)
  Resize((480, 480), keep_ratio=False)                                    es_resize_u8 on the device (pipeline.py)
  RandomFlip3D + GlobalRotScaleTrans (transforms/augmentation.py:         draw_augmentation(): the draws; points are
      87-139, 322-447)                                                    transformed inside es_depth_to_points, the
                                                                          boxes by pipeline.augment_gt_boxes

The draws are taken from one numpy RandomState in the reference's order, so a seeded run of the reference pipeline and of
this one take the same decisions (pinned for the view choice and PointSample by tests/golden/dataset_parse.pkl)."""
import threading

import numpy as np


def decode_image(path, out=None):
    """-> (H, W, 3) uint8 RGB.  (The reference decodes to BGR with cv2 and the data preprocessor swaps to RGB,
    data_preprocessor.py `bgr_to_rgb=True`; PIL yields RGB directly.)  out: destination of that shape (a frame of the worker's
    shared slot): the decoded pixels are written there instead of into a fresh array."""
    from PIL import Image
    with Image.open(path) as im:
        a = np.asarray(im if im.mode == 'RGB' else im.convert('RGB'))       # (convert() of an RGB image is a full copy)
    if out is None or out.shape != a.shape:       # (a frame of another shape comes back as its own array: the pipeline's
        return a                                   #  uniformity check reports it)
    out[...] = a
    return out


_HOST_LIB = False          # libes_host.so (csrc/host_codec.c, include/es_host.h): False = not looked for yet, None = absent


def _host_lib():
    global _HOST_LIB
    if _HOST_LIB is False:
        import ctypes
        import os
        path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'libes_host.so')
        try:
            lib = ctypes.CDLL(path)
            lib.es_png_gray16_to_f32.restype = ctypes.c_int
            lib.es_png_gray16_to_f32.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_void_p]
            _HOST_LIB = lib
        except OSError:
            _HOST_LIB = None                  # (the generic decoder below yields the same values, more slowly)
    return _HOST_LIB


_DEFLATE_LIB = False       # libdeflate.so.0: False = not looked for yet, None = absent
_DEFLATE_TLS = threading.local()    # per thread (and per forked worker): its own decompressor + scratch buffers -- the loader's
                                    # thread workers decode concurrently (ctypes releases the GIL) and a decompressor is not re-entrant


def _inflate(z, n):
    """zlib stream -> exactly n bytes (a ctypes buffer or bytes), or None when the stream is broken / of another length.
    libdeflate (present in the ROCm image as a system library: whole-buffer inflate, 2-3 x zlib on depth maps: 2.2 -> 1.1 ms
    for a 640x480 frame) when it can be loaded, python's zlib otherwise -- inflate is deterministic, the bytes are the same.
    The returned buffer is this thread's scratch: valid until its next call."""
    global _DEFLATE_LIB
    import os
    if _DEFLATE_LIB is False:
        try:
            import ctypes
            lib = ctypes.CDLL('libdeflate.so.0')
            lib.libdeflate_alloc_decompressor.restype = ctypes.c_void_p
            lib.libdeflate_zlib_decompress.restype = ctypes.c_int
            lib.libdeflate_zlib_decompress.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_void_p,
                                                       ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t)]
            _DEFLATE_LIB = lib
        except (OSError, AttributeError):
            _DEFLATE_LIB = None
    st = None
    if _DEFLATE_LIB is not None:
        st = getattr(_DEFLATE_TLS, 'state', None)
        if st is None or st[2] != os.getpid():
            dec = _DEFLATE_LIB.libdeflate_alloc_decompressor()
            st = _DEFLATE_TLS.state = (_DEFLATE_LIB, dec, os.getpid(), {}) if dec else None
    if st is not None:
        import ctypes
        lib, dec, _, scratch = st
        buf = scratch.get(n)
        if buf is None:
            if len(scratch) > 4:
                scratch.clear()
            buf = scratch[n] = ctypes.create_string_buffer(n)
        got = ctypes.c_size_t(0)
        if lib.libdeflate_zlib_decompress(dec, z, len(z), buf, n, ctypes.byref(got)) == 0 and got.value == n:
            return buf
        return None
    import zlib
    try:
        raw = zlib.decompress(z)
    except zlib.error:
        return None
    return raw if len(raw) == n else None


def _inflate_gray16_png(data):
    """PNG file bytes -> (H, W, inflated scanlines) when the file is a non-interlaced 16-bit greyscale PNG with intact
    chunk CRCs, else None (the caller then takes the generic decoder, which also owns the error reporting)."""
    import struct
    import zlib
    if data[:8] != b'\x89PNG\r\n\x1a\n':
        return None
    o, n, shape, idat, ended = 8, len(data), None, [], False
    while o + 12 <= n:
        ln = int.from_bytes(data[o:o + 4], 'big')
        ty = data[o + 4:o + 8]
        if o + 12 + ln > n:
            return None
        body = data[o + 8:o + 8 + ln]
        if zlib.crc32(body, zlib.crc32(ty)) != int.from_bytes(data[o + 8 + ln:o + 12 + ln], 'big'):
            return None
        if ty == b'IHDR':
            if ln != 13 or shape is not None:
                return None
            w, h, depth, colour, comp, filt, lace = struct.unpack('>IIBBBBB', body)
            if (depth, colour, comp, filt, lace) != (16, 0, 0, 0, 0) or w == 0 or h == 0:
                return None
            shape = (h, w)
        elif ty == b'IDAT':
            idat.append(body)
        elif ty == b'IEND':
            ended = True
            break
        o += 12 + ln
    if shape is None or not idat or not ended:
        return None
    raw = _inflate(b''.join(idat) if len(idat) > 1 else idat[0], shape[0] * (1 + 2 * shape[1]))
    if raw is None:
        return None
    return shape[0], shape[1], raw


def decode_depth(path, depth_shift, out=None):
    """16-bit PNG -> float32 metres: `imfrombytes(flag='unchanged').astype(float32) / depth_shift` (loading.py:68-73).
    out: float32 destination (see decode_image); a map of another shape comes back as its own array.
    16-bit greyscale PNGs -- the dataset's depth format -- take the native one-pass path (csrc/host_codec.c: unfilter + byte
    swap + float conversion + division straight into `out`; zlib inflates); every other file, and every file that path
    declines, goes through PIL with the same f32 arithmetic (the samples are cast to float32, then divided in float32)."""
    lib = _host_lib()
    if lib is not None:
        with open(path, 'rb') as f:
            data = f.read()
        got = _inflate_gray16_png(data)
        if got is not None:
            h, w, raw = got
            dst = out if out is not None and out.shape == (h, w) and out.dtype == np.float32 and out.flags.c_contiguous \
                else np.empty((h, w), np.float32)
            if lib.es_png_gray16_to_f32(raw, h, w, float(np.float32(depth_shift)), dst.ctypes.data) == 0:
                return dst
    from PIL import Image
    with Image.open(path) as im:
        a = np.asarray(im)
    if a.ndim != 2:
        raise ValueError(f'{path}: depth image must have one channel, got shape {a.shape}')
    if out is None or out.shape != a.shape:
        return a.astype(np.float32) / np.float32(depth_shift)
    np.divide(a, np.float32(depth_shift), out=out, dtype=np.float32, casting='unsafe')
    return out


def frame_shape(path):
    """(H, W) of an image file from its header (no decode)"""
    from PIL import Image
    with Image.open(path) as im:
        w, h = im.size
    return h, w


def select_views(n_total, n_images, ordered, rng):
    """ids of the frames MultiViewPipeline keeps (multiview.py:46-64, including the ordered-mode stride rule)"""
    ids = np.arange(n_total)
    replace = n_images > len(ids)
    if ordered:
        step = (len(ids) - 1) // (n_images - 1)
        if step > 0:
            return ids[::step][:n_images]
        return rng.choice(ids, n_images, replace=replace)
    return rng.choice(ids, n_images, replace=replace)


def draw_without_order(rng, n, k, exact=True):
    """k of range(n) (with replacement iff n < k), the reference's `np.random.choice(range(n), k, replace=n < k)`.
    exact: the legacy RandomState call itself -- the SAME stream as the reference, but without replacement it permutes all n
    (a 300 k-element Fisher-Yates per depth frame: 52 % of the host time of one scan, profiles/r3_loader_profile.txt).
    not exact: the same distribution from an O(k) draw (numpy Generator, Floyd's algorithm) seeded by ONE integer taken from
    `rng`, so a seeded run stays reproducible -- it just no longer replays the reference's individual picks."""
    if exact or n < k:
        return rng.choice(n, k, replace=n < k)
    gen = np.random.Generator(np.random.PCG64(int(rng.randint(0, 2 ** 31 - 1))))
    return gen.choice(n, k, replace=False, shuffle=False)


def sample_pixels(depth, num_points, rng, exact=True):
    """PointSample on the points of one frame: the points are the non-zero depth pixels in raster order
    (points.py:46-53), `choice(range(len), num, replace = len < num)` (points.py:189-206).  Returns pixel indices;
    an all-zero depth map yields no points (points.py:132-134)."""
    nz = np.flatnonzero(depth.reshape(-1))
    if len(nz) == 0:
        return nz.astype(np.int32)
    # (an int population draws the same stream as the reference's `range(len(points))`, without materialising it)
    choices = draw_without_order(rng, len(nz), num_points, exact)
    return nz[choices].astype(np.int32)


def points_in_range(depths, depth_intr, extr, sel_view, sel_pix, point_range):
    """mask over the chosen pixels: is the aggregated GLOBAL point inside `point_range` (BasePoints.in_range_3d: strict
    inequalities on all six faces)?  Same f32 arithmetic as ConvertRGBDToPoints + AggregateMultiViewPoints
    (points.py:46-53, structures/bbox_3d/utils.py:357-366, multiview.py:151-153): (u d, v d, d, 1) inv(pad4(K))^T, then the
    camera->global transform; only the chosen pixels are un-projected."""
    keep = np.zeros(len(sel_pix), bool)
    lo, hi = np.asarray(point_range[:3], np.float32), np.asarray(point_range[3:], np.float32)
    for v in range(len(depths)):
        m = np.flatnonzero(sel_view == v)
        if len(m) == 0:
            continue
        d = depths[v]
        W = d.shape[1]
        pix = sel_pix[m]
        dd = d.reshape(-1)[pix].astype(np.float32)
        u, w_ = (pix % W).astype(np.float32), (pix // W).astype(np.float32)
        K = np.eye(4, dtype=np.float32)
        k = np.asarray(depth_intr[v], np.float32)
        K[:k.shape[0], :k.shape[1]] = k
        homo = np.stack([u * dd, w_ * dd, dd, np.ones_like(dd)], 1)
        cam = (homo @ np.linalg.inv(K).T.astype(np.float32))[:, :3]
        E = np.linalg.inv(np.asarray(extr[v], np.float32)).astype(np.float32)
        g = cam @ E[:3, :3].T + E[:3, 3]
        keep[m] = np.all((g > lo) & (g < hi), axis=1)
    return keep


def draw_augmentation(cfg, rng):
    """RandomFlip3D then GlobalRotScaleTrans: two rand() for the flips (augmentation.py:114-123), uniform rotation
    (negated, 377-382), uniform scale (445-446), normal translation (360-361).  -> the `aug` dict of pipeline.py"""
    hf = bool(rng.rand() < cfg['flip_h']) if cfg.get('flip') else False
    vf = bool(rng.rand() < cfg['flip_v']) if cfg.get('flip') else False
    aug = dict(hflip=hf, vflip=vf, rot=np.eye(3, dtype=np.float32), scale=1.0, trans=np.zeros(3, np.float32))
    meta = dict(transformation_3d_flow=(['HF'] if hf else []) + (['VF'] if vf else []))
    if cfg.get('flip'):
        meta.update(pcd_horizontal_flip=hf, pcd_vertical_flip=vf)
    if cfg.get('rst'):
        ang = -rng.uniform(cfg['rot_range'][0], cfg['rot_range'][1])
        scale = float(rng.uniform(cfg['scale_range'][0], cfg['scale_range'][1]))
        trans = rng.normal(scale=np.array(cfg['trans_std'], dtype=np.float32), size=3).T.astype(np.float32)
        c, s = np.cos(ang), np.sin(ang)
        # rotation about z applied as points @ rot_mat_T (base_points.py:198-201, euler_box3d.py rotate)
        rot_mat_T = np.array([[c, s, 0], [-s, c, 0], [0, 0, 1]], dtype=np.float32)
        aug.update(rot=rot_mat_T, scale=scale, trans=trans)
        meta.update(pcd_rotation=rot_mat_T, pcd_rotation_angle=ang, pcd_scale_factor=scale, pcd_trans=trans)
        meta['transformation_3d_flow'] += ['R', 'S', 'T']
    return aug, meta


class ScanPipeline:
    """The parameters of a reference pipeline config (list of transform dicts) + the host work it implies."""

    def __init__(self, n_images=20, ordered=False, n_points=100000, view_points=None, img_scale=(480, 480),
                 flip=False, flip_h=0.5, flip_v=0.5, rst=False, rot_range=(-0.087266, 0.087266), scale_range=(.9, 1.1),
                 trans_std=(.1, .1, .1), with_occupancy=False, view_masks=False, point_range=None, exact_draws=True,
                 device_draws=False):
        # exact_draws: PointSample replays the reference's RandomState stream pick for pick (parity tests, golden vectors);
        # False: same distribution from O(k) draws (draw_without_order) -- 1.7 x the scans per core
        self.exact_draws = bool(exact_draws)
        # device_draws (round 4, N4): the host only DECODES; both PointSample draws are made on the GPU from counter-based keys
        # (pipeline.device_point_sample: the same law, neither numpy stream).  Used when no `replace=True` case of the reference
        # arises (every frame holds >= view_points valid pixels, V * view_points >= n_points) and no range filter sits between
        # the two draws; otherwise the scan falls back to the host's O(k) draws.  Overrides exact_draws.
        self.device_draws = bool(device_draws)
        self.n_images, self.ordered = n_images, ordered
        self.n_points, self.view_points = n_points, view_points if view_points is not None else n_points // 10
        self.img_scale = tuple(img_scale)                      # (w, h) as in mmcv Resize
        self.aug = dict(flip=flip, flip_h=flip_h, flip_v=flip_v, rst=rst, rot_range=tuple(rot_range),
                        scale_range=tuple(scale_range), trans_std=tuple(trans_std))
        self.with_occupancy, self.view_masks = with_occupancy, view_masks
        # PointsRangeFilter (occupancy configs): the reference filters the aggregated cloud on the CPU and THEN samples
        # n_points among the survivors (transforms/points.py:232-262); here the n_points pixels are chosen first (the
        # host never sees 3-D coordinates) and the device drops the out-of-range ones when it voxelises -- the cloud is
        # thinner by the out-of-range fraction.  Kept in the scan so the consumer can apply it.
        self.point_range = None if point_range is None else tuple(point_range)

    @classmethod
    def from_cfg(cls, pipeline):
        """read the knobs out of the reference's transform list; unknown transforms are an error, not a silent skip"""
        kw = {}
        if isinstance(pipeline, ScanPipeline):
            return pipeline
        for t in pipeline or ():
            ty = t['type'].split('.')[-1]
            if ty == 'MultiViewPipeline':
                kw.update(n_images=t['n_images'], ordered=t.get('ordered', False))
                for u in t['transforms']:
                    uy = u['type'].split('.')[-1]
                    if uy == 'PointSample':
                        kw['view_points'] = u['num_points']
                    elif uy == 'Resize':
                        assert not u.get('keep_ratio', False), 'Resize(keep_ratio=True) is not used by the shipped configs'
                        kw['img_scale'] = tuple(u['scale'])
                    elif uy == 'ConvertRGBDToPoints':
                        assert not u.get('use_color', False), 'coloured points are not part of the device path'
                    elif uy not in ('LoadImageFromFile', 'LoadDepthFromFile'):
                        raise NotImplementedError(f'per-view transform {uy}')
            elif ty == 'PointSample':
                kw['n_points'] = t['num_points']
            elif ty == 'RandomFlip3D':
                kw.update(flip=True, flip_h=t.get('flip_ratio_bev_horizontal', 0.0), flip_v=t.get('flip_ratio_bev_vertical', 0.0))
            elif ty == 'GlobalRotScaleTrans':
                std = t.get('translation_std', (0, 0, 0))
                kw.update(rst=True, rot_range=tuple(t.get('rot_range', (-0.78539816, 0.78539816))),
                          scale_range=tuple(t.get('scale_ratio_range', (0.95, 1.05))),
                          trans_std=tuple(std) if isinstance(std, (list, tuple)) else (std,) * 3)
            elif ty == 'LoadAnnotations3D':
                kw['with_occupancy'] = bool(t.get('with_occupancy', False))
            elif ty == 'PointsRangeFilter':
                kw['point_range'] = tuple(t['point_cloud_range'])
            elif ty == 'ConstructMultiViewMasks':
                kw['view_masks'] = True
            elif ty in ('AggregateMultiViewPoints', 'Pack3DDetInputs'):
                pass
            else:
                raise NotImplementedError(f'pipeline transform {ty}')
        return cls(**kw)

    def __call__(self, info, rng, alloc=None):
        """info: one parsed data_info -> raw scan dict (numpy), the exchange format of pipeline.pin_scan/upload_scan:
        depth (V,H,W) f32 metres, img_raw (V,H,W,3) u8 RGB at the file's resolution (resized on the device),
        extrinsic / intrinsic (V,4,4), sel_view / sel_pix (n_points,), gt_boxes (G,9) augmented, gt_labels, meta, aug.
        alloc(V, (H, W), (Hd, Wd)) -> (img_raw buffer (V,H,W,3) u8, depth buffer (V,Hd,Wd) f32) or None: where the frames are
        decoded INTO (a loader worker hands out views of its shared slot: no stack copy and no slot copy afterwards --
        2 x 43 MB of memcpy per scan at the shipped sizes); default: fresh arrays, still without the stack copy."""
        ids = select_views(len(info['img_path']), self.n_images, self.ordered, rng)
        depths, imgs, sel_view, sel_pix = [], [], [], []
        intr_all, extr = info['depth2img']['intrinsic'], []
        intr = []
        dci = info['depth_cam2img']
        depth_intr = []
        buf_i = buf_d = None
        in_place = True                          # every frame landed in the buffers (they ARE the stacks then)
        for j, i in enumerate(ids.tolist()):
            if j == 0:                           # frame sizes from the first headers; every kind must be uniform inside a scan
                ish, dsh = frame_shape(info['img_path'][i]), frame_shape(info['depth_img_path'][i])
                bufs = alloc(len(ids), ish, dsh) if alloc is not None else None
                buf_i, buf_d = bufs if bufs is not None else (np.empty((len(ids),) + ish + (3,), np.uint8),
                                                              np.empty((len(ids),) + dsh, np.float32))
            vi, vd = buf_i[j], buf_d[j]
            im = decode_image(info['img_path'][i], vi)
            d = decode_depth(info['depth_img_path'][i], info['depth_shift'], vd)
            in_place = in_place and im is vi and d is vd
            imgs.append(im)
            depths.append(d)
            if not self.device_draws:
                pix = sample_pixels(d, self.view_points, rng, self.exact_draws)
                sel_pix.append(pix)
                sel_view.append(np.full(len(pix), j, np.int32))
            intr.append(np.asarray(intr_all[i] if isinstance(intr_all, list) else intr_all, np.float32))
            depth_intr.append(np.asarray(dci[i] if isinstance(dci, list) else dci, np.float32))
            extr.append(info['depth2img']['extrinsic'][i])
        on_device = False
        if self.device_draws:
            # the device draws cover the `replace=False` law only: enough valid pixels in every frame, enough aggregated points,
            # no PointsRangeFilter between the draws -- else the host draws (O(k)) take over for this scan
            on_device = (self.point_range is None and len(depths) * self.view_points >= self.n_points and
                         len(depths) <= 256 and all(int(np.count_nonzero(d != 0)) >= self.view_points for d in depths))   # (14 x faster than count_nonzero(d) on f32)
            if not on_device:
                for j, d in enumerate(depths):
                    pix = sample_pixels(d, self.view_points, rng, False)
                    sel_pix.append(pix)
                    sel_view.append(np.full(len(pix), j, np.int32))
        if not on_device:
            sel_view, sel_pix = np.concatenate(sel_view), np.concatenate(sel_pix)
        draw_seed = int(rng.randint(0, 2 ** 31 - 1)) if on_device else None
        # colour frames and depth maps come at their own native resolutions (e.g. ScanNet 1296x968 jpg, 640x480 png: the
        # reference keeps a separate depth_cam2img for that reason); each kind must be uniform inside a scan because the
        # frames of a kind are stacked, but the two kinds are independent -- depth_to_points uses the depth size, the frame
        # resize the colour size
        ishapes, dshapes = {im.shape[:2] for im in imgs}, {d.shape for d in depths}
        if len(ishapes) != 1 or len(dshapes) != 1:
            raise ValueError(f"{info['sample_idx']}: the colour frames of one scan must share a resolution and so must its "
                             f"depth maps, got colour {sorted(ishapes)} depth {sorted(dshapes)}")
        # PointsRangeFilter sits BETWEEN the aggregation and PointSample(n_points) in the occupancy pipeline
        # (configs/occupancy/mv-occ_...py:121-123, transforms/points.py:246-263): drop the aggregated points outside the
        # range (unless fewer than 100 survive), THEN draw -- so the draw's population is the filtered cloud
        if self.point_range is not None and not on_device and len(sel_pix):
            keep = points_in_range(depths, depth_intr, extr, sel_view, sel_pix, self.point_range)
            if int(keep.sum()) >= 100:
                sel_view, sel_pix = sel_view[keep], sel_pix[keep]
        # PointSample(n_points) over the aggregated cloud (points.py:189-206)
        if not on_device and len(sel_pix):
            pick = draw_without_order(rng, len(sel_pix), self.n_points, self.exact_draws and not self.device_draws)
            sel_view, sel_pix = sel_view[pick], sel_pix[pick]
        aug, aug_meta = draw_augmentation(self.aug, rng)
        H, W = imgs[0].shape[:2]
        w_new, h_new = self.img_scale
        ann = info.get('ann_info') or info.get('eval_ann_info') or {}
        boxes = np.asarray(ann.get('gt_bboxes_3d', np.zeros((0, 9), np.float32)), np.float32)
        labels = np.asarray(ann.get('gt_labels_3d', np.zeros((0,), np.int64)), np.int64)
        from ..pipeline import augment_gt_boxes
        meta = dict(depth2img=dict(extrinsic=extr, intrinsic=intr, origin=info['depth2img']['origin']),
                    img_shape=(h_new, w_new), ori_shape=(H, W), scale_factor=(w_new / W, h_new / H),
                    box_type_3d='euler-depth', scan_id=info['scan_id'], sample_idx=info['sample_idx'],
                    img_path=[info['img_path'][i] for i in ids.tolist()])
        # the remaining keys Pack3DDetInputs copies into the data sample when the pipeline produced them
        # (transforms/formatting.py:66-79)
        for k in ('axis_align_matrix', 'cam2img'):
            if k in info:
                meta[k] = info[k]
        meta.update(aug_meta)
        scan = dict(depth=buf_d if in_place else np.stack(depths), img_raw=buf_i if in_place else np.stack(imgs), extrinsic=np.stack(extr).astype(np.float32),
                    intrinsic=np.stack(depth_intr), gt_boxes=augment_gt_boxes(boxes, aug).numpy(), gt_labels=labels, meta=meta, aug=aug)
        if on_device:
            scan['draw'] = (draw_seed, int(self.view_points), int(self.n_points))
        else:
            scan.update(sel_view=sel_view.astype(np.int32), sel_pix=sel_pix.astype(np.int32))
        if self.with_occupancy and 'gt_occupancy' in ann:
            scan['gt_occupancy'] = ann['gt_occupancy']
            vm = ann.get('visible_occupancy_masks')
            if vm is not None and len(vm) and len(vm[0]):
                vm = [np.asarray(vm[i]) for i in ids.tolist()]
                scan['visible_occupancy_masks'] = vm
                if self.view_masks:
                    # ConstructMultiViewMasks (multiview.py:253-268): OR over the chosen frames -- the reference's loop
                    # is `range(1, len(img) - 1)`, i.e. the LAST frame never contributes; kept as is
                    m = vm[0]
                    for j in range(1, len(vm) - 1):
                        m = np.logical_or(m, vm[j])
                    scan['gt_occupancy_masks'] = m
        if self.point_range is not None:
            scan['point_range'] = self.point_range
        if 'visible_instance_masks' in ann:
            scan['visible_instance_masks'] = [ann['visible_instance_masks'][i] for i in ids.tolist()]
        return scan
