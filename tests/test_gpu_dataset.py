"""SURVEY N4 on the GPU: the device half of the real-data path (frame resize, unprojection of decoded depth) against the
oracle, and a train step fed from files through the threaded loader against the oracle's losses."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'these tests need the MI355X'
    return torch.device('cuda:0')


def test_resize_kernel_bit_exact(dev):
    """es_resize_u8 == oracle/resize.py (OpenCV's 8-bit INTER_LINEAR rule) on down-, same- and up-scaling, every byte"""
    from embodiedscan_amd import pipeline
    from oracle.resize import resize_u8
    rng = np.random.default_rng(1)
    for (H, W), (h, w), V in (((480, 640), (480, 480), 3), ((60, 80), (48, 48), 5), ((37, 53), (64, 96), 2),
                              ((480, 480), (480, 480), 1)):
        img = rng.integers(0, 256, (V, H, W, 3), dtype=np.uint8)
        got = pipeline.resize_frames(torch.from_numpy(img).to(dev), (h, w)).cpu().numpy()
        want = resize_u8(img, (h, w)).transpose(0, 3, 1, 2)
        assert got.shape == want.shape == (V, 3, h, w)
        assert np.array_equal(got, want), (H, W, h, w, int(np.abs(got.astype(int) - want.astype(int)).max()))


def _pipe(n_images, view_points, n_points, scale):
    return [dict(type='LoadAnnotations3D'),
            dict(type='MultiViewPipeline', n_images=n_images,
                 transforms=[dict(type='LoadImageFromFile'), dict(type='LoadDepthFromFile'),
                             dict(type='ConvertRGBDToPoints', coord_type='CAMERA'),
                             dict(type='PointSample', num_points=view_points),
                             dict(type='Resize', scale=scale, keep_ratio=False)]),
            dict(type='AggregateMultiViewPoints', coord_type='DEPTH'), dict(type='PointSample', num_points=n_points),
            dict(type='RandomFlip3D', sync_2d=False, flip_2d=False, flip_ratio_bev_horizontal=0.5, flip_ratio_bev_vertical=0.5),
            dict(type='GlobalRotScaleTrans', rot_range=[-0.087266, 0.087266], scale_ratio_range=[.9, 1.1],
                 translation_std=[.1, .1, .1], shift_height=False),
            dict(type='Pack3DDetInputs', keys=['img', 'points', 'gt_bboxes_3d', 'gt_labels_3d'])]


def test_train_step_from_files_matches_oracle(dev, tmp_path):
    """files -> threaded loader (pinned) -> async copy + device resize -> A1-A3 -> detector.train_step, against the oracle
    fed with the same decoded arrays: points 2e-5 m, frames bit-exact, the three losses 2e-3 relative (f32 mode)"""
    from embodiedscan_amd import engine as E, pipeline, synth
    from embodiedscan_amd.config import build_detector, build_optim_wrapper, load_config
    from embodiedscan_amd.datasets import EmbodiedScanDataset, ScanLoader
    from oracle import model as OM, pipeline as OP
    from oracle.resize import resize_u8
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    names = [f'class{i}' for i in range(284)]
    synth.write_dataset(str(tmp_path), n_scans=3, n_frames=5, height=120, width=160, n_boxes=6, class_names=names, seed=3,
                        occupancy=False)
    # 'scannet' scans without occupancy files would fail to parse: this dataset is detection-only -> arkitscenes-style ids
    import pickle
    p = os.path.join(str(tmp_path), 'embodiedscan_infos_train.pkl')
    with open(p, 'rb') as f:
        ann = pickle.load(f)
    for d in ann['data_list']:
        d['sample_idx'] = d['sample_idx'].replace('scannet/', 'arkitscenes/')
    with open(p, 'wb') as f:
        pickle.dump(ann, f)
    ds = EmbodiedScanDataset(str(tmp_path), 'embodiedscan_infos_train.pkl', metainfo=dict(classes=names),
                             pipeline=_pipe(3, 2500, 6000, (128, 128)), remove_dontcare=True, filter_empty_gt=False)
    assert len(ds) == 3
    loader = ScanLoader(ds, batch_size=2, shuffle=False, num_threads=3, pin=True)
    batch_pinned = next(iter(loader))
    assert all(s['img_raw'].is_pinned() and s['depth'].is_pinned() for s in batch_pinned)
    copy_stream = torch.cuda.Stream()
    slots = [pipeline.alloc_slot(s, dev) for s in batch_pinned]
    with torch.cuda.stream(copy_stream):
        dscans = [pipeline.upload_into(sl, s) for sl, s in zip(slots, batch_pinned)]
    torch.cuda.current_stream().wait_stream(copy_stream)
    # the same scans decoded again on the host for the oracle (the loader's decisions are a function of seed/position)
    raw = [ds.load_scan(i, np.random.RandomState((0 * 1000003 + 0 * 7919 + pos) % (2 ** 32))) for pos, i in enumerate((0, 1))]
    cfg = load_config(os.path.join(root, 'configs', 'mv_3ddet.py'))
    old = E.PRECISION[0]
    E.PRECISION[0] = 'f32'
    try:
        det = build_detector(cfg, device=dev, seed=0).to(dev)
        batch = pipeline.make_batch(dscans)
        sd = {k: v.cpu() for k, v in det.state_dict().items()}
        pts = [q.cpu() for q in batch['inputs']['points']]
        for q, r in zip(pts, raw):
            err = float((q - OP.scan_to_points(r)).abs().max())
            print(f'A1-A3 on decoded depth: max abs err {err:.2e} m (tol 2e-5)')
            assert err < 2e-5
        frames = [resize_u8(r['img_raw'], (128, 128)).transpose(0, 3, 1, 2) for r in raw]
        assert all(np.array_equal(d['img'].cpu().numpy(), f) for d, f in zip(dscans, frames))
        losses = det.train_step(batch, build_optim_wrapper(cfg))
        torch.cuda.synchronize()
    finally:
        E.PRECISION[0] = old
    imgs = torch.stack([OM.preprocess_img(torch.from_numpy(f), [123.675, 116.28, 103.53], [58.395, 57.12, 57.375]) for f in frames])
    ol = OM.detector_loss(sd, pts, imgs, [r['meta'] for r in raw], [torch.from_numpy(r['gt_boxes']) for r in raw],
                          [torch.from_numpy(r['gt_labels']) for r in raw])
    for k in ol:
        a, b = float(losses[k]), float(ol[k])
        print(f'real-data step {k}: hip {a:.6f} oracle {b:.6f}')
        assert abs(a - b) <= 2e-3 * max(abs(b), 1e-3), (k, a, b)


def test_grounding_step_from_files(dev, tmp_path):
    """MultiView3DGroundingDataset -> process loader (shared pinned slots) -> grounder.train_step: finite losses, and the
    batch built from the loader's scans equals the one built from explicitly passed annotations"""
    from embodiedscan_amd import engine as E, pipeline, synth
    from embodiedscan_amd.config import build_detector, build_optim_wrapper, load_config
    from embodiedscan_amd.datasets import MultiView3DGroundingDataset, ScanLoader
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    synth.write_dataset(str(tmp_path), n_scans=2, n_frames=5, height=120, width=160, n_boxes=6, seed=4, n_voxels=(8, 8, 4))
    pipe = _pipe(3, 2500, 6000, (128, 128))
    pipe = pipe[:4] + pipe[5:]                                  # grounding config: no RandomFlip3D
    ds = MultiView3DGroundingDataset(str(tmp_path), 'embodiedscan_infos_train.pkl', 'embodiedscan_train_vg.json',
                                     metainfo=dict(classes='all'), pipeline=pipe, tokens_positive_rebuild=True)
    keep = [i for i in range(len(ds)) if 'tokens_positive' in ds.get_data_info(i)]
    ds.data_list = [ds.data_list[i] for i in keep]
    loader = ScanLoader(ds, batch_size=2, shuffle=False, num_threads=2, prefetch=2, pin=True, workers='process')
    batch_pinned = next(iter(loader))
    assert all(s['depth'].is_pinned() for s in batch_pinned)
    slots = [pipeline.alloc_slot(s, dev) for s in batch_pinned]
    dscans = [pipeline.upload_into(sl, s) for sl, s in zip(slots, batch_pinned)]
    ev = torch.cuda.Event()
    ev.record()
    loader.done(batch_pinned, ev)
    loader.close()
    cfg = load_config(os.path.join(root, 'configs', 'mv_grounding.py'))
    det = build_detector(cfg, device=dev, seed=0).to(dev)
    data = pipeline.make_grounding_batch(dscans)
    anns = [dict(text=d['text'], tokens_positive=d['tokens_positive'], gt_boxes=d['gt_boxes'], gt_labels=d['gt_labels'])
            for d in dscans]
    data2 = pipeline.make_grounding_batch(dscans, anns)
    for a, b in zip(data['data_samples'], data2['data_samples']):
        assert a.text == b.text and a.tokens_positive == b.tokens_positive
        assert torch.equal(a.gt_instances_3d.bboxes_3d.tensor, b.gt_instances_3d.bboxes_3d.tensor)
    losses = det.train_step(data, build_optim_wrapper(cfg))
    torch.cuda.synchronize()
    vals = {k: float(v) for k, v in losses.items()}
    print('grounding step from files:', {k: round(v, 4) for k, v in list(vals.items())[:4]}, '...')
    assert len(vals) >= 2 and all(np.isfinite(v) for v in vals.values())
