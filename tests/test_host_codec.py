"""The loader's native depth decoder (embodiedscan_amd/csrc/host_codec.c through datasets/loading.decode_depth; SURVEY N4,
reference transforms/loading.py:68-73) against PIL on 16-bit greyscale PNGs written here with EVERY PNG filter type (one type
per file, and a different type on every row), split IDAT chunks and odd sizes -- bit-identical float32 metres; files the
native path must decline (8-bit, interlaced, corrupted chunk) take the generic decoder with its values / its error."""
import struct
import zlib

import numpy as np
import pytest


def _filter_rows(img16, types):
    """(H, W) uint16 -> PNG scanlines (filter byte + filtered big-endian samples) with filter type types[r] on row r,
    straight from the PNG specification (section 9: Sub / Up / Average / Paeth on bytes, 2 bytes per pixel)"""
    H, W = img16.shape
    be = img16.astype('>u2').view(np.uint8).reshape(H, 2 * W).astype(np.int32)
    out = bytearray()
    prev = np.zeros(2 * W, np.int32)
    for r in range(H):
        cur = be[r]
        left = np.concatenate([[0, 0], cur[:-2]])
        ul = np.concatenate([[0, 0], prev[:-2]])
        t = int(types[r])
        if t == 0:
            f = cur
        elif t == 1:
            f = cur - left
        elif t == 2:
            f = cur - prev
        elif t == 3:
            f = cur - ((left + prev) >> 1)
        else:
            p = left + prev - ul
            pa, pb, pc = np.abs(p - left), np.abs(p - prev), np.abs(p - ul)
            f = cur - np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, prev, ul))
        out.append(t)
        out += (f & 255).astype(np.uint8).tobytes()
        prev = cur
    return bytes(out)


def _chunk(ty, body):
    return struct.pack('>I', len(body)) + ty + body + struct.pack('>I', zlib.crc32(body, zlib.crc32(ty)))


def _write_png16(path, img16, types, idat_split=1, extra=b''):
    H, W = img16.shape
    z = zlib.compress(_filter_rows(img16, types), 6)
    parts = [z[i * len(z) // idat_split:(i + 1) * len(z) // idat_split] for i in range(idat_split)]
    with open(path, 'wb') as f:
        f.write(b'\x89PNG\r\n\x1a\n' + _chunk(b'IHDR', struct.pack('>IIBBBBB', W, H, 16, 0, 0, 0, 0)) + extra +
                b''.join(_chunk(b'IDAT', p) for p in parts) + _chunk(b'IEND', b''))


def _depth_like(rng, H, W):
    y, x = np.mgrid[0:H, 0:W]
    d = 1500 + 900 * np.sin(x / 37.0) * np.cos(y / 23.0) + rng.integers(0, 9, (H, W))
    d[rng.random((H, W)) < 0.07] = 0                       # invalid pixels
    d[:, W // 3:W // 3 + 2] = 65535                        # extremes: wrap-around of every byte filter
    return d.clip(0, 65535).astype(np.uint16)


def _lib():
    """libes_host.so is a build product (git-ignored): a fresh checkout compiles it here (gcc, one second)"""
    import os
    import subprocess
    from embodiedscan_amd.datasets import loading as L
    if L._host_lib() is None:
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        subprocess.check_call(['make', '-C', os.path.join(root, 'embodiedscan_amd', 'csrc'), '../libes_host.so'])
        L._HOST_LIB = False
    assert L._host_lib() is not None, 'libes_host.so missing: run `make -C embodiedscan_amd/csrc` (__graft_entry__.build)'
    return L._host_lib()


def _pil(path, shift):
    from PIL import Image
    with Image.open(path) as im:
        return np.asarray(im).astype(np.float32) / np.float32(shift)


def test_native_depth_decoder_is_loaded_and_equals_pil_for_every_filter_type(tmp_path):
    from embodiedscan_amd.datasets import loading as L
    _lib()
    rng = np.random.default_rng(3)
    n = 0
    for H, W in ((48, 64), (37, 53), (1, 1), (2, 300), (120, 160), (5, 2), (4, 3), (3, 1), (7, 4)):
        img = _depth_like(rng, H, W) if H > 2 and W > 8 else rng.integers(0, 65536, (H, W)).astype(np.uint16)
        cases = [[t] * H for t in range(5)] + [list(rng.integers(0, 5, H)) for _ in range(3)]
        for types in cases:
            for split in (1, 3):
                p = str(tmp_path / 'd.png')
                _write_png16(p, img, types, idat_split=split)
                want = _pil(p, 1000.0)
                assert np.array_equal(want, img.astype(np.float32) / np.float32(1000.0))      # (the writer itself is right)
                got = L.decode_depth(p, 1000.0)
                assert got.dtype == np.float32 and got.shape == (H, W) and np.array_equal(got, want), (H, W, types[:4], split)
                out = np.full((H, W), -1, np.float32)
                assert L.decode_depth(p, 4000, out) is out and np.array_equal(out, _pil(p, 4000))
                n += 1
    # a destination of another shape is left alone and the map comes back as its own array
    other = np.full((3, 3), -1, np.float32)
    got = L.decode_depth(p, 1000.0, other)
    assert got is not other and np.array_equal(got, want) and float(other.min()) == -1
    print(f'native depth decoder: {n} files identical to PIL (5 filter types, mixed rows, split IDAT, odd sizes)')


def test_native_depth_decoder_declines_what_it_does_not_own(tmp_path, monkeypatch):
    from PIL import Image
    from embodiedscan_amd.datasets import loading as L
    calls = []
    lib = _lib()
    real = lib.es_png_gray16_to_f32

    class Spy:
        def __getattr__(self, k):
            return getattr(lib, k)

        def es_png_gray16_to_f32(self, *a):
            calls.append(1)
            return real(*a)
    monkeypatch.setattr(L, '_HOST_LIB', Spy())
    rng = np.random.default_rng(5)
    img = _depth_like(rng, 40, 56)
    # ancillary chunks before the data are skipped, the native path still runs
    p = str(tmp_path / 'a.png')
    _write_png16(p, img, [4] * 40, extra=_chunk(b'tEXt', b'Software\x00synthetic') + _chunk(b'pHYs', struct.pack('>IIB', 1, 1, 0)))
    assert np.array_equal(L.decode_depth(p, 1000.0), _pil(p, 1000.0)) and len(calls) == 1
    # 8-bit greyscale: PIL's values, native path not taken
    p8 = str(tmp_path / 'g8.png')
    Image.fromarray((img >> 8).astype(np.uint8)).save(p8)
    assert np.array_equal(L.decode_depth(p8, 10.0), (img >> 8).astype(np.float32) / np.float32(10.0)) and len(calls) == 1
    # PIL-written 16-bit file (its own filter choices / zlib settings)
    p16 = str(tmp_path / 'pil16.png')
    Image.fromarray(img).save(p16)
    assert np.array_equal(L.decode_depth(p16, 1000.0), img.astype(np.float32) / np.float32(1000.0)) and len(calls) == 2
    # a corrupted data chunk (CRC mismatch): the generic decoder reports it, as before
    data = bytearray(open(p, 'rb').read())
    at = data.index(b'IDAT') + 40
    data[at] ^= 0x55
    pc = str(tmp_path / 'corrupt.png')
    open(pc, 'wb').write(bytes(data))
    with pytest.raises(Exception):
        L.decode_depth(pc, 1000.0)
    assert len(calls) == 2
    # three-channel "depth" image: the documented ValueError
    prgb = str(tmp_path / 'rgb.png')
    Image.fromarray(np.zeros((8, 8, 3), np.uint8)).save(prgb)
    with pytest.raises(ValueError, match='one channel'):
        L.decode_depth(prgb, 1000.0)


def test_scan_pipeline_is_identical_with_and_without_the_native_decoder(tmp_path, monkeypatch):
    """one scan of a written dataset through ScanPipeline: every array of the scan dict is the same with libes_host.so and
    with the PIL path (the loader's workers may run either)"""
    from embodiedscan_amd import synth
    from embodiedscan_amd.datasets import EmbodiedScanDataset
    from embodiedscan_amd.datasets import loading as L
    names = [f'class{i}' for i in range(12)]
    root = str(tmp_path / 'ds')
    synth.write_dataset(root, n_scans=1, n_frames=5, height=60, width=80, n_boxes=4, class_names=names, seed=2)
    pipe = [dict(type='LoadAnnotations3D'),
            dict(type='MultiViewPipeline', n_images=4, transforms=[dict(type='LoadImageFromFile'), dict(type='LoadDepthFromFile'),
                                                                  dict(type='ConvertRGBDToPoints', coord_type='CAMERA'),
                                                                  dict(type='PointSample', num_points=200),
                                                                  dict(type='Resize', scale=(64, 64), keep_ratio=False)]),
            dict(type='AggregateMultiViewPoints', coord_type='DEPTH'), dict(type='PointSample', num_points=600),
            dict(type='Pack3DDetInputs', keys=['img', 'points', 'gt_bboxes_3d', 'gt_labels_3d'])]
    ds = EmbodiedScanDataset(root, 'embodiedscan_infos_train.pkl', metainfo=dict(classes=names), pipeline=pipe)
    _lib()
    a = ds.load_scan(0, np.random.RandomState(4))
    monkeypatch.setattr(L, '_HOST_LIB', None)
    b = ds.load_scan(0, np.random.RandomState(4))
    for k in ('depth', 'img_raw', 'sel_view', 'sel_pix', 'extrinsic', 'intrinsic', 'gt_boxes'):
        assert np.array_equal(a[k], b[k]), k
    assert a['depth'].dtype == np.float32 and a['depth'].flags.c_contiguous


def test_depth_decoder_same_values_through_libdeflate_zlib_and_from_many_threads(tmp_path, monkeypatch):
    """the inflate step runs through libdeflate when the system has it, python's zlib otherwise; the loader's thread workers
    decode concurrently (per-thread decompressor and scratch) -- always the same float32 maps"""
    import threading
    from embodiedscan_amd.datasets import loading as L
    _lib()
    rng = np.random.default_rng(11)
    files = []
    for i, (H, W) in enumerate(((96, 128), (96, 128), (50, 70), (120, 160))):
        img = _depth_like(rng, H, W)
        p = str(tmp_path / f'd{i}.png')
        _write_png16(p, img, list(rng.integers(0, 5, H)), idat_split=1 + i % 3)
        files.append((p, img.astype(np.float32) / np.float32(1000.0)))
    with_deflate = [L.decode_depth(p, 1000.0) for p, _ in files]
    has = L._DEFLATE_LIB is not None
    for (p, want), got in zip(files, with_deflate):
        assert np.array_equal(got, want)
    errors = []

    def work(seed):
        order = np.random.default_rng(seed).permutation(len(files) * 25) % len(files)
        for k in order:
            p, want = files[k]
            if not np.array_equal(L.decode_depth(p, 1000.0), want):
                errors.append((seed, p))
    threads = [threading.Thread(target=work, args=(s,)) for s in range(8)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:3]
    monkeypatch.setattr(L, '_DEFLATE_LIB', None)                     # the zlib fallback
    for p, want in files:
        assert np.array_equal(L.decode_depth(p, 1000.0), want)
    # a truncated stream is declined by either inflater (the generic decoder then reports the file)
    import zlib
    z = zlib.compress(b'x' * 1000)
    assert L._inflate(z, 999) is None and L._inflate(z[:-6], 1000) is None
    monkeypatch.setattr(L, '_DEFLATE_LIB', False)
    assert L._inflate(z, 999) is None and L._inflate(z[:-6], 1000) is None
    assert bytes(L._inflate(z, 1000)[:1000]) == b'x' * 1000
    print('libdeflate present:', has)
