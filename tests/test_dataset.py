"""the `.pkl` annotation reader and the host half of the real-data path against golden vectors made by
the reference's own EmbodiedScanDataset / MultiViewPipeline / PointSample (oracle/make_golden_dataset.py), plus round
trips through a freshly written synthetic dataset and the DefaultSampler-style rank sharding."""
import os
import pickle

import numpy as np
import pytest

FIX_NAMES = None


def _names():
    from embodiedscan_amd.synth import _NOUNS
    return _NOUNS + ['object']


def _same(a, b, path=''):
    """recursive equality of parsed infos (arrays exact, golden paths carry '<root>')"""
    if isinstance(b, dict):
        assert isinstance(a, dict), path
        kb = set(b)
        ka = set(a) - {'box_type_3d'}
        assert ka == kb, (path, sorted(ka ^ kb))
        for k in kb:
            _same(a[k], b[k], f'{path}.{k}')
    elif isinstance(b, (list, tuple)):
        assert len(a) == len(b), (path, len(a), len(b))
        for i, (x, y) in enumerate(zip(a, b)):
            _same(x, y, f'{path}[{i}]')
    elif isinstance(b, np.ndarray):
        a = np.asarray(a)
        assert a.shape == b.shape and a.dtype == b.dtype, (path, a.shape, b.shape, a.dtype, b.dtype)
        assert np.array_equal(a, b), path
    elif isinstance(b, str):
        assert a == b, (path, a, b)
    else:
        assert a == b, (path, a, b)


def _strip_root(x, root):
    if isinstance(x, dict):
        return {k: _strip_root(v, root) for k, v in x.items()}
    if isinstance(x, list):
        return [_strip_root(v, root) for v in x]
    if isinstance(x, str):
        return x.replace(root, '<root>')
    return x


@pytest.fixture(scope='module')
def golden(golden_dir):
    with open(os.path.join(golden_dir, 'dataset_parse.pkl'), 'rb') as f:
        return pickle.load(f)


@pytest.mark.parametrize('tag', ['train', 'test', 'subset_nodontcare', 'default_classes'])
def test_reader_matches_reference_parse(golden, golden_dir, tag):
    """every key of every parsed info (paths, extrinsics = inv(axis_align @ cam2global), intrinsics, depth_shift, boxes,
    mapped labels, visibility masks, occupancy with the 255 rule, eval_ann_info) equals the reference's, bit for bit"""
    from embodiedscan_amd.datasets import EmbodiedScanDataset
    names = _names()
    kw = dict(train=dict(metainfo=dict(classes=names)), test=dict(metainfo=dict(classes=names), test_mode=True),
              subset_nodontcare=dict(metainfo=dict(classes=names[:5], occ_classes=names[:7]), remove_dontcare=True),
              default_classes=dict(metainfo=dict(occ_classes=names)))[tag]
    root = os.path.join(golden_dir, 'fake_dataset')
    ds = EmbodiedScanDataset(data_root=root, ann_file='embodiedscan_infos_train.pkl', pipeline=[], **kw)
    g = golden[tag]
    assert list(ds.metainfo['classes']) == g['classes']
    assert np.array_equal(ds.label_mapping, g['label_mapping']) and np.array_equal(ds.occ_label_mapping, g['occ_label_mapping'])
    assert len(ds) == len(g['data_list']) == 2
    _same(_strip_root(ds.data_list, root), g['data_list'], tag)


def test_view_choice_and_point_sample_follow_the_reference_stream(golden):
    """same seeded numpy stream -> same frames (incl. the ordered-mode stride rule and its fall-back) and same PointSample
    choices as the reference transforms"""
    from embodiedscan_amd.datasets.loading import sample_pixels, select_views
    for v in golden['views']:
        ids = select_views(v['n_total'], v['n_images'], v['ordered'], np.random.RandomState(v['seed']))
        assert np.array_equal(ids, v['ids']), v
    for s in golden['point_sample']:
        depth = np.ones(s['n'], np.float32)                      # every pixel valid: pixel index == point index
        pix = sample_pixels(depth, s['num'], np.random.RandomState(s['seed']))
        assert np.array_equal(pix, s['choices']), (s['seed'], s['n'])
    # zero-depth pixels are not points; an all-zero frame contributes nothing
    d = np.zeros((4, 5), np.float32)
    d[1, 2] = d[3, 4] = 1.0
    assert set(sample_pixels(d, 10, np.random.RandomState(0)).tolist()) == {7, 19}
    assert len(sample_pixels(np.zeros((4, 5), np.float32), 10, np.random.RandomState(0))) == 0


PIPE = [dict(type='LoadAnnotations3D'),
        dict(type='MultiViewPipeline', n_images=4,
             transforms=[dict(type='LoadImageFromFile'), dict(type='LoadDepthFromFile'),
                         dict(type='ConvertRGBDToPoints', coord_type='CAMERA'), dict(type='PointSample', num_points=300),
                         dict(type='Resize', scale=(48, 48), keep_ratio=False)]),
        dict(type='AggregateMultiViewPoints', coord_type='DEPTH'), dict(type='PointSample', num_points=1000),
        dict(type='RandomFlip3D', sync_2d=False, flip_2d=False, flip_ratio_bev_horizontal=0.5, flip_ratio_bev_vertical=0.5),
        dict(type='GlobalRotScaleTrans', rot_range=[-0.087266, 0.087266], scale_ratio_range=[.9, 1.1],
             translation_std=[.1, .1, .1], shift_height=False),
        dict(type='Pack3DDetInputs', keys=['img', 'points', 'gt_bboxes_3d', 'gt_labels_3d'])]


def test_written_dataset_round_trips(tmp_path):
    """synthetic scan -> files -> reader: depth exact to the PNG's millimetre grid, camera matrices to f32 rounding,
    JPEG frames within codec error, boxes = augment_gt_boxes(raw), decisions reproducible from the RandomState"""
    from embodiedscan_amd import synth
    from embodiedscan_amd.datasets import EmbodiedScanDataset
    from embodiedscan_amd.pipeline import augment_gt_boxes
    src, names = synth.write_dataset(str(tmp_path), n_scans=2, n_frames=6, n_voxels=(8, 8, 4), seed=5)
    ds = EmbodiedScanDataset(str(tmp_path), 'embodiedscan_infos_train.pkl', metainfo=dict(classes=names), pipeline=PIPE)
    assert ds.pipeline.n_images == 4 and ds.pipeline.view_points == 300 and ds.pipeline.n_points == 1000
    assert ds.pipeline.img_scale == (48, 48) and ds.pipeline.aug['flip'] and ds.pipeline.aug['rst']
    a = ds.load_scan(1, np.random.RandomState(3))
    b = ds.load_scan(1, np.random.RandomState(3))
    for k in ('depth', 'img_raw', 'sel_view', 'sel_pix', 'gt_boxes', 'extrinsic'):
        assert np.array_equal(a[k], b[k]), k
    ids = [int(os.path.basename(p)[:5]) for p in a['meta']['img_path']]
    s = src[1]
    assert a['depth'].shape == (4, 60, 80) and a['img_raw'].shape == (4, 60, 80, 3) and a['img_raw'].dtype == np.uint8
    assert np.array_equal(a['depth'], (np.rint(s['depth'][ids] * 1000.0) / np.float32(1000.0)).astype(np.float32))
    assert np.abs(a['extrinsic'] - s['extrinsic'][ids]).max() < 2e-6
    assert np.array_equal(a['intrinsic'], s['intrinsic'][ids])
    assert a['meta']['img_shape'] == (48, 48) and a['meta']['scale_factor'] == (48 / 80, 48 / 60)
    assert len(a['sel_pix']) == 1000 and a['sel_view'].max() < 4 and (a['depth'].reshape(4, -1)[a['sel_view'], a['sel_pix']] > 0).all()
    raw = ds.get_data_info(1)['ann_info']['gt_bboxes_3d']
    assert np.allclose(raw, s['gt_boxes'], atol=1e-6)
    assert np.array_equal(a['gt_boxes'], augment_gt_boxes(raw, a['aug']).numpy())
    # meta keys of the data sample (the subset of Pack3DDetInputs.meta_keys this pipeline produces)
    assert {'img_path', 'ori_shape', 'img_shape', 'depth2img', 'scale_factor', 'pcd_horizontal_flip', 'pcd_vertical_flip',
            'box_type_3d', 'pcd_trans', 'sample_idx', 'pcd_scale_factor', 'pcd_rotation', 'pcd_rotation_angle',
            'transformation_3d_flow', 'axis_align_matrix', 'cam2img', 'scan_id'} <= set(a['meta'])
    flow = a['meta']['transformation_3d_flow']
    assert flow[-3:] == ['R', 'S', 'T'] and ('HF' in flow) == a['aug']['hflip'] and ('VF' in flow) == a['aug']['vflip']
    # unknown transforms are refused, not skipped
    with pytest.raises(NotImplementedError):
        EmbodiedScanDataset(str(tmp_path), 'embodiedscan_infos_train.pkl', metainfo=dict(classes=names),
                            pipeline=PIPE + [dict(type='PointShuffle')])


def test_rank_sharding_is_default_sampler(tmp_path):
    """shards: equal length on every rank, union covers the (repeated) dataset, identical to a literal restatement of
    mmengine's DefaultSampler (seeded randperm, wrap-around padding, rank::world)"""
    import math
    import torch
    from embodiedscan_amd.datasets import shard_indices
    for n, world, times in ((10, 4, 1), (7, 8, 1), (5, 2, 10), (16, 4, 1)):
        for epoch in (0, 3):
            g = torch.Generator()
            g.manual_seed(11 + epoch)
            perm = torch.randperm(n * times, generator=g).tolist()
            size = math.ceil(n * times / world) * world
            padded = (perm * int(size / len(perm) + 1))[:size]
            shards = [shard_indices(n, r, world, True, 11, epoch, True, times) for r in range(world)]
            assert all(s == [i % n for i in padded[r:size:world]] for r, s in enumerate(shards))
            assert len({len(s) for s in shards}) == 1
            assert set(sum(shards, [])) == set(range(n))
    assert shard_indices(5, 1, 2, shuffle=False) == [1, 3, 0]


def test_loader_is_ordered_and_thread_count_independent(tmp_path):
    from embodiedscan_amd import synth
    from embodiedscan_amd.datasets import EmbodiedScanDataset, ScanLoader
    _, names = synth.write_dataset(str(tmp_path), n_scans=3, n_frames=5, n_voxels=(8, 8, 4), seed=2)
    ds = EmbodiedScanDataset(str(tmp_path), 'embodiedscan_infos_train.pkl', metainfo=dict(classes=names), pipeline=PIPE)
    runs = []
    for threads in (1, 4):
        ld = ScanLoader(ds, batch_size=2, rank=1, world=2, seed=4, times=4, num_threads=threads, prefetch=3, pin=False)
        assert len(ld) == 3
        runs.append([[(s['meta']['scan_id'], s['sel_pix'].clone(), s['depth'].clone()) for s in b] for b in ld])
    want = [ds.get_data_info(i)['scan_id'] for i in ScanLoader(ds, 2, 1, 2, seed=4, times=4).indices()]
    assert [s[0] for b in runs[0] for s in b] == want[:6]
    for ba, bb in zip(*runs):
        for (ia, pa, da), (ib, pb, db) in zip(ba, bb):
            assert ia == ib and bool((pa == pb).all()) and bool((da == db).all())
    # a failing decode surfaces in the consumer
    os.remove(ds.get_data_info(0)['depth_img_path'][0])
    os.remove(ds.get_data_info(1)['depth_img_path'][0])
    os.remove(ds.get_data_info(2)['depth_img_path'][0])
    ds.pipeline.n_images = 5                                     # every frame is read -> the missing one is hit
    with pytest.raises(Exception):
        list(ScanLoader(ds, batch_size=1, num_threads=2, pin=False))


def test_resize_oracle_properties():
    from oracle.resize import linear_tables, resize_u8
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (2, 30, 40, 3), dtype=np.uint8)
    assert np.array_equal(resize_u8(img, (30, 40)), img)                          # identity tables: (2048, 0)
    o, c = linear_tables(640, 480)
    assert o.min() == 0 and o.max() <= 639 and (c.sum(1) == 2048).all() and c.min() >= 0
    up = resize_u8(np.full((1, 4, 4, 3), 77, np.uint8), (9, 11))
    assert up.shape == (1, 9, 11, 3) and (up == 77).all()                          # constants are preserved
    # product tables == oracle tables
    from embodiedscan_amd.pipeline import _axis_table
    for a, b in ((640, 480), (480, 480), (80, 48), (60, 100)):
        po, pc = _axis_table(a, b)
        oo, oc = linear_tables(a, b)
        assert np.array_equal(po.numpy(), oo) and np.array_equal(pc.numpy(), oc)


def test_occupancy_pipeline_fields(tmp_path):
    """occupancy config knobs: LoadAnnotations3D(with_occupancy), PointsRangeFilter, ConstructMultiViewMasks (OR over
    the chosen frames except the last one, as the reference's loop does)"""
    from embodiedscan_amd import synth
    from embodiedscan_amd.datasets import EmbodiedScanDataset
    _, names = synth.write_dataset(str(tmp_path), n_scans=1, n_frames=6, n_voxels=(8, 8, 4), seed=9)
    pipe = [dict(type='LoadAnnotations3D', with_occupancy=True, with_visible_occupancy_masks=True)] + PIPE[1:3] + [
        dict(type='PointsRangeFilter', point_cloud_range=[-3.2, -3.2, -1.28, 3.2, 3.2, 1.28]),
        dict(type='PointSample', num_points=1000), dict(type='ConstructMultiViewMasks'),
        dict(type='Pack3DDetInputs', keys=['img', 'points', 'gt_bboxes_3d', 'gt_labels_3d', 'gt_occupancy'])]
    ds = EmbodiedScanDataset(str(tmp_path), 'embodiedscan_infos_train.pkl', metainfo=dict(classes=names, occ_classes=names),
                             pipeline=pipe)
    sc = ds.load_scan(0, np.random.RandomState(1))
    ann = ds.get_data_info(0)['ann_info']
    assert np.array_equal(sc['gt_occupancy'], ann['gt_occupancy']) and sc['gt_occupancy'][:, 3].max() == 255
    ids = [int(os.path.basename(p)[:5]) for p in sc['meta']['img_path']]
    want = ann['visible_occupancy_masks'][ids[0]]
    for i in ids[1:-1]:
        want = np.logical_or(want, ann['visible_occupancy_masks'][i])
    assert sc['gt_occupancy_masks'].shape == (8, 8, 4) and np.array_equal(sc['gt_occupancy_masks'], want)
    assert sc['point_range'] == (-3.2, -3.2, -1.28, 3.2, 3.2, 1.28) and not sc['aug']['hflip'] and sc['aug']['scale'] == 1.0


def test_build_dataloader_from_config(tmp_path):
    """the data section of configs/mv_3ddet.py (= the reference config's, :134-200): RepeatDataset -> times,
    DefaultSampler -> shard, pipeline knobs (20 frames, 100k points, 480x480)"""
    from embodiedscan_amd import synth
    from embodiedscan_amd.config import build_dataloader
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    _, names = synth.write_dataset(str(tmp_path), n_scans=2, n_frames=5, n_voxels=(8, 8, 4))
    ds, ld = build_dataloader(os.path.join(root, 'configs', 'mv_3ddet.py'), data_root=str(tmp_path),
                              metainfo=dict(classes=names), pin=False, rank=1, world=2)
    p = ds.pipeline
    assert (p.n_images, p.view_points, p.n_points, p.img_scale, p.ordered) == (20, 10000, 100000, (480, 480), False)
    assert ld.times == 10 and ld.batch_size == 4 and ld.shuffle and not ld.drop_last and len(ld) == 3   # ceil(20 scans / 2 ranks / 4)
    batch = next(iter(ld))
    assert len(batch) == 4 and batch[0]['img_raw'].shape == (20, 60, 80, 3) and batch[0]['sel_pix'].numel() == 100000
    assert batch[0]['meta']['img_shape'] == (480, 480)
    _, val = build_dataloader(os.path.join(root, 'configs', 'mv_3ddet.py'), split='val', data_root=str(tmp_path),
                              ann_file='embodiedscan_infos_train.pkl', metainfo=dict(classes=names), pin=False)
    assert val.dataset.test_mode and val.dataset.pipeline.ordered and val.dataset.pipeline.n_images == 50 and not val.shuffle
    assert 'eval_ann_info' in val.dataset.get_data_info(0)


def test_process_loader_matches_thread_loader(tmp_path):
    """forked workers writing into shared slots: same scans, same bytes, same order as the threaded loader; slots are
    recycled through done(); a consumer that never releases a batch gets a loud error instead of a silent overwrite"""
    from embodiedscan_amd import synth
    from embodiedscan_amd.datasets import EmbodiedScanDataset, ScanLoader
    _, names = synth.write_dataset(str(tmp_path), n_scans=3, n_frames=5, n_voxels=(8, 8, 4), seed=2)
    ds = EmbodiedScanDataset(str(tmp_path), 'embodiedscan_infos_train.pkl', metainfo=dict(classes=names), pipeline=PIPE)
    ref = [[{k: (v.clone() if hasattr(v, 'clone') else v) for k, v in s.items()} for s in b]
           for b in ScanLoader(ds, batch_size=2, seed=4, times=6, num_threads=2, prefetch=3, pin=False)]
    ld = ScanLoader(ds, batch_size=2, seed=4, times=6, num_threads=3, prefetch=3, pin=False, workers='process')
    n = 0
    for got, want in zip(ld, ref):
        for a, b in zip(got, want):
            assert a['meta']['scan_id'] == b['meta']['scan_id'] and a['meta']['img_path'] == b['meta']['img_path']
            for k in ('depth', 'img_raw', 'sel_view', 'sel_pix', 'mats', 'aug', 'gt_boxes', 'gt_labels'):
                assert a[k].dtype == b[k].dtype and a[k].shape == b[k].shape and bool((a[k] == b[k]).all()), k
        ld.done(got)
        n += 1
    assert n == len(ref) == 9                                    # 18 scans through 7 slots: recycling works
    ld.close()
    ld2 = ScanLoader(ds, batch_size=2, seed=4, times=6, num_threads=2, prefetch=1, pin=False, workers='process')
    with pytest.raises(RuntimeError, match='done'):
        held = []
        for batch in ld2:
            held.append(batch)                                   # never released
    ld2.close()
    assert ld2._pending == [] and ld2._free == [] and ld2._slabs is None
    # a re-iteration after close() starts from a clean slot table (no stale ids from the abandoned epoch)
    first = next(iter(ld2))
    assert first[0]['meta']['scan_id'] == ref[0][0]['meta']['scan_id'] and sorted(ld2._free + first.slots) == list(range(len(ld2._slabs)))
    ld2.close()


def test_process_loader_mixed_resolutions_and_dead_workers(tmp_path):
    """EmbodiedScan mixes ScanNet / 3RScan / Matterport3D frame sizes: the shared slots are sized from the frame headers of
    every source; a scan that still does not fit (slot_bytes forced small here) is decoded by the parent instead of
    aborting the epoch; a worker that dies raises instead of hanging the consumer (round-2 advisor findings)"""
    import os as _os
    import pickle
    import signal
    from embodiedscan_amd import synth
    from embodiedscan_amd.datasets import EmbodiedScanDataset, ScanLoader
    _, names = synth.write_dataset(str(tmp_path), n_scans=2, n_frames=5, height=60, width=80, n_voxels=(8, 8, 4), seed=2)
    big = tmp_path / 'big'
    synth.write_dataset(str(big), n_scans=1, n_frames=5, height=120, width=160, n_voxels=(8, 8, 4), seed=3)
    # graft the larger scan into the first dataset as a second "source"
    with open(tmp_path / 'embodiedscan_infos_train.pkl', 'rb') as f:
        a = pickle.load(f)
    with open(big / 'embodiedscan_infos_train.pkl', 'rb') as f:
        b = pickle.load(f)
    e = b['data_list'][0]
    import shutil
    name = e['sample_idx'].split('/')[1]
    e['sample_idx'] = '3rscan/' + name
    shutil.copytree(str(big / 'scannet' / 'scans' / name / 'occupancy'), str(tmp_path / '3rscan' / name / 'occupancy'))
    for im in e['images']:
        im['img_path'] = _os.path.join('big', im['img_path'])
        im['depth_path'] = _os.path.join('big', im['depth_path'])
    a['data_list'].append(e)
    with open(tmp_path / 'mixed.pkl', 'wb') as f:
        pickle.dump(a, f)
    ds = EmbodiedScanDataset(str(tmp_path), 'mixed.pkl', metainfo=dict(classes=names), pipeline=PIPE)
    want = [ds.get_data_info(i)['sample_idx'] for i in range(len(ds))]
    if not any(w.startswith('3rscan') for w in want):
        pytest.skip('the reader dropped the grafted scan')
    ld = ScanLoader(ds, batch_size=1, shuffle=False, num_threads=2, prefetch=2, pin=False, workers='process')
    shapes = []
    for batch in ld:
        shapes.append(tuple(batch[0]['img_raw'].shape[1:3]))
        ld.done(batch)
    assert (120, 160) in shapes and (60, 80) in shapes           # the header probe sized the slots for the large source
    small_need = ld._slot_bytes
    ld.close()
    # forced-small slots: the large scan takes the parent's slow path, same bytes
    ld = ScanLoader(ds, batch_size=1, shuffle=False, num_threads=2, prefetch=2, pin=False, workers='process',
                    slot_bytes=small_need // 3)
    got = []
    for batch in ld:
        got.append((tuple(batch[0]['img_raw'].shape[1:3]), int(batch[0]['img_raw'].long().sum())))
        ld.done(batch)
    ld.close()
    ref = [(tuple(b[0]['img_raw'].shape[1:3]), int(b[0]['img_raw'].long().sum()))
           for b in ScanLoader(ds, batch_size=1, shuffle=False, num_threads=1, pin=False)]
    assert got == ref
    # a dead worker is noticed
    ld = ScanLoader(ds, batch_size=1, shuffle=False, times=50, num_threads=1, prefetch=1, pin=False, workers='process',
                    worker_timeout=1.0)
    it = iter(ld)
    ld.done(next(it))
    _os.kill(ld._procs[0].pid, signal.SIGKILL)
    with pytest.raises(RuntimeError, match='died'):
        for _ in range(20):
            ld.done(next(it))
    ld.close()


@pytest.mark.parametrize('tag', ['vg_train', 'vg_test'])
def test_grounding_reader_matches_reference(golden, golden_dir, tag):
    """MultiView3DGroundingDataset: one sample per language annotation -- targets looked up by bbox_id (single, list,
    missing -> dropped, none -> all boxes), tokens_positive given or rebuilt from the target phrase, hard / unique /
    view-dependent flags -- equal to the reference's class key for key"""
    from embodiedscan_amd.datasets import MultiView3DGroundingDataset
    names = _names()
    kw = dict(vg_train=dict(metainfo=dict(classes=names), tokens_positive_rebuild=True),
              vg_test=dict(metainfo=dict(classes=names), test_mode=True, tokens_positive_rebuild=False))[tag]
    root = os.path.join(golden_dir, 'fake_dataset')
    ds = MultiView3DGroundingDataset(data_root=root, ann_file='embodiedscan_infos_train.pkl',
                                     vg_file='embodiedscan_train_vg.json', pipeline=[], **kw)
    g = golden[tag]
    assert np.array_equal(ds.label_mapping, g['label_mapping']) and len(ds) == len(g['data_list']) == 6
    _same(_strip_root(ds.data_list, root), g['data_list'], tag)


def test_grounding_scan_from_files(tmp_path):
    from embodiedscan_amd import synth
    from embodiedscan_amd.datasets import MultiView3DGroundingDataset
    from embodiedscan_amd.pipeline import augment_gt_boxes
    src, names = synth.write_dataset(str(tmp_path), n_scans=1, n_frames=5, n_voxels=(8, 8, 4), seed=7)
    pipe = PIPE[:4] + [PIPE[5], PIPE[6]]                       # the grounding config has no RandomFlip3D
    ds = MultiView3DGroundingDataset(str(tmp_path), 'embodiedscan_infos_train.pkl', 'embodiedscan_train_vg.json',
                                     metainfo=dict(classes='all'), pipeline=pipe, tokens_positive_rebuild=True)
    assert len(ds) == 3 and not ds.pipeline.aug['flip'] and ds.pipeline.aug['rst']
    sc = ds.load_scan(0, np.random.RandomState(0))
    noun = sc['text'].split()[2]
    assert sc['text'].startswith('find the') and sc['tokens_positive'] == [[[9, 9 + len(noun)]]]
    assert sc['gt_boxes'].shape == (1, 9) and not sc['aug']['hflip'] and sc['meta']['is_view_dep'] and sc['meta']['is_hard']
    assert np.allclose(sc['gt_boxes'], augment_gt_boxes(src[0]['gt_boxes'][:1], sc['aug']).numpy(), atol=1e-6)
    multi = ds.load_scan(1, np.random.RandomState(0))
    assert multi['gt_boxes'].shape == (2, 9) and multi['tokens_positive'] == [[[8, 14]], [[8, 14]]] and multi['meta']['is_unique']
    every = ds.load_scan(2, np.random.RandomState(0))
    assert every['gt_boxes'].shape == (6, 9) and 'tokens_positive' not in every


def test_depth_and_colour_resolutions_are_independent(tmp_path):
    """real scans store colour and depth at different native sizes (ScanNet 1296x968 jpg vs 640x480 png, hence the
    reference's separate depth_cam2img): the file-backed path must load them, un-project with the DEPTH size / intrinsics and
    keep ori_shape / scale_factor from the COLOUR frames (round-2 advisor finding: it raised)"""
    import torch
    from embodiedscan_amd import synth
    from embodiedscan_amd.datasets import EmbodiedScanDataset
    from oracle import pipeline as OP
    srcs, names = synth.write_dataset(str(tmp_path), n_scans=1, n_frames=4, height=60, width=80, seed=3, depth_div=2)
    ds = EmbodiedScanDataset(str(tmp_path), 'embodiedscan_infos_train.pkl', metainfo=dict(classes=names), pipeline=PIPE)
    sc = ds.load_scan(0, np.random.RandomState(0))
    assert sc['img_raw'].shape[1:3] == (60, 80) and sc['depth'].shape[1:] == (30, 40)
    assert sc['meta']['ori_shape'] == (60, 80)
    assert int(sc['sel_pix'].max()) < 30 * 40
    # the un-projected cloud lies on the source scan's surfaces: every point reproduces a pixel of the FULL-resolution depth
    pts = OP.scan_to_points(dict(sc, aug=dict(hflip=False, vflip=False, rot=np.eye(3, dtype=np.float32), scale=1.0,
                                              trans=np.zeros(3, np.float32))))
    assert torch.isfinite(pts).all() and pts.shape == (len(sc['sel_pix']), 3)
    full = srcs[0]['depth']
    v, p = sc['sel_view'], sc['sel_pix']
    ids = [int(os.path.basename(q)[:5]) for q in sc['meta']['img_path']]
    want = np.array([full[ids[a], 2 * (b // 40), 2 * (b % 40)] for a, b in zip(v.tolist(), p.tolist())])
    got = sc['depth'].reshape(len(ids), -1)[v, p]
    assert np.abs(got - want).max() < 1e-3          # millimetre PNG quantisation
    # mixed sizes inside one kind are still refused
    from PIL import Image
    f = os.path.join(str(tmp_path), sc['meta']['img_path'][0].replace('.jpg', '.png')) if not os.path.isabs(sc['meta']['img_path'][0]) \
        else sc['meta']['img_path'][0].replace('.jpg', '.png')
    Image.fromarray(np.zeros((10, 10), np.uint16)).save(f)
    with pytest.raises(ValueError, match='share a resolution'):
        ds.load_scan(0, np.random.RandomState(0))


def test_points_range_filter_precedes_point_sample(tmp_path):
    """occupancy pipeline order (configs/occupancy/mv-occ_...py:121-123): aggregate -> PointsRangeFilter -> PointSample.
    Every point the loader hands over lies strictly inside the range, and the draw runs over the FILTERED population: with
    a range that keeps only part of the cloud, PointSample(n) still returns n points, all of them inside."""
    import torch
    from embodiedscan_amd import synth
    from embodiedscan_amd.datasets import EmbodiedScanDataset
    from oracle import pipeline as OP
    _, names = synth.write_dataset(str(tmp_path), n_scans=1, n_frames=6, n_voxels=(8, 8, 4), seed=9)
    rng_box = [-2.0, -1.5, -0.5, 3.2, 1.5, 2.0]
    pipe = [dict(type='LoadAnnotations3D', with_occupancy=True, with_visible_occupancy_masks=True)] + PIPE[1:3] + [
        dict(type='PointsRangeFilter', point_cloud_range=rng_box),
        dict(type='PointSample', num_points=1000), dict(type='ConstructMultiViewMasks'),
        dict(type='Pack3DDetInputs', keys=['img', 'points', 'gt_bboxes_3d', 'gt_labels_3d', 'gt_occupancy'])]
    ds = EmbodiedScanDataset(str(tmp_path), 'embodiedscan_infos_train.pkl', metainfo=dict(classes=names, occ_classes=names),
                             pipeline=pipe)
    sc = ds.load_scan(0, np.random.RandomState(1))
    pts = OP.scan_to_points(sc).numpy()
    assert pts.shape == (1000, 3)
    lo, hi = np.array(rng_box[:3], np.float32), np.array(rng_box[3:], np.float32)
    assert ((pts > lo) & (pts < hi)).all()
    # without the filter the same scan has points outside that box (the filter did something)
    ds2 = EmbodiedScanDataset(str(tmp_path), 'embodiedscan_infos_train.pkl', metainfo=dict(classes=names, occ_classes=names),
                              pipeline=[pipe[0]] + PIPE[1:3] + pipe[4:])
    p2 = OP.scan_to_points(ds2.load_scan(0, np.random.RandomState(1))).numpy()
    assert not ((p2 > lo) & (p2 < hi)).all()


def test_fast_point_sample_draws_same_law_not_same_stream():
    """ScanPipeline(exact_draws=False): PointSample by an O(k) draw instead of the legacy RandomState permutation of the whole
    population (loading.draw_without_order).  Same law -- k distinct members of range(n), every member equally likely,
    with replacement iff n < k -- reproducible from the seed, and only non-zero depth pixels are ever chosen; the exact mode
    stays the reference's stream (the test above)."""
    import numpy as np
    from embodiedscan_amd.datasets.loading import draw_without_order, sample_pixels
    a = draw_without_order(np.random.RandomState(3), 300000, 10000, exact=False)
    b = draw_without_order(np.random.RandomState(3), 300000, 10000, exact=False)
    assert np.array_equal(a, b) and len(np.unique(a)) == 10000 and a.min() >= 0 and a.max() < 300000
    assert not np.array_equal(a, np.random.RandomState(3).choice(300000, 10000, replace=False))     # a different stream
    assert np.array_equal(draw_without_order(np.random.RandomState(3), 300000, 10000, exact=True),
                          np.random.RandomState(3).choice(300000, 10000, replace=False))
    # n < k: with replacement, the legacy call in both modes
    assert np.array_equal(draw_without_order(np.random.RandomState(4), 50, 200, exact=False),
                          np.random.RandomState(4).choice(50, 200, replace=True))
    # uniformity: 400 draws of 1000 from 20000 -> every decile of the population receives 10 % +- 0.5 %
    rng = np.random.RandomState(5)
    hist = np.zeros(10)
    for _ in range(400):
        hist += np.bincount(draw_without_order(rng, 20000, 1000, exact=False) // 2000, minlength=10)
    assert np.all(np.abs(hist / hist.sum() - 0.1) < 0.005), hist / hist.sum()
    depth = np.zeros((48, 64), np.float32)
    depth[10:30, 5:50] = 1.5
    pix = sample_pixels(depth, 300, np.random.RandomState(6), exact=False)
    assert len(pix) == 300 and len(np.unique(pix)) == 300 and np.all(depth.reshape(-1)[pix] > 0)
