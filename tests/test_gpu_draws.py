"""N4: PointSample on the device (pipeline.device_point_sample, csrc/data.hip es_draw_*) against its integer-for-integer CPU
restatement (oracle/draws.py, whose LAW is tested in tests/test_draws.py): the selected (view, pixel) lists must be IDENTICAL,
element by element, including the order; plus a train step fed by the loader with device_draws=True."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize('V,H,W,vp,npts', [(3, 48, 64, 300, 700), (20, 120, 160, 2000, 15000), (35, 60, 80, 500, 9000),
                                            (20, 480, 640, 10000, 100000)])
def test_device_point_sample_equals_the_restatement(V, H, W, vp, npts):
    from embodiedscan_amd import pipeline
    from oracle import draws as D
    dev = torch.device('cuda:0')
    rng = np.random.default_rng(V * 1000 + H)
    depth = (rng.random((V, H, W)) > 0.25).astype(np.float32) * (0.5 + rng.random((V, H, W)).astype(np.float32))
    depth[0, :, : W // 3] = 0                                  # a view with a large hole
    for seed in (1, 2 ** 31 - 5):
        sv, sp = pipeline.device_point_sample(torch.from_numpy(depth).to(dev), seed, vp, npts)
        torch.cuda.synchronize()
        ev, ep = D.point_sample(depth, seed, vp, npts)
        np.testing.assert_array_equal(sv.cpu().numpy(), ev)
        np.testing.assert_array_equal(sp.cpu().numpy(), ep)
    assert (depth.reshape(V, -1)[ev, ep] != 0).all()
    print(f'device PointSample {V} x {H}x{W}: {vp} per view -> {npts}: identical to oracle/draws.py for two seeds')


def test_train_step_from_files_with_device_draws(tmp_path):
    """EmbodiedScanDataset -> ScanLoader(device_draws=True) -> upload_batch (draws on the upload stream) -> train step"""
    from embodiedscan_amd import engine as E, pipeline, synth
    from embodiedscan_amd.config import build_detector, build_optim_wrapper, load_config
    from embodiedscan_amd.datasets import EmbodiedScanDataset, ScanLoader
    dev = torch.device('cuda:0')
    root = str(tmp_path)
    names = [f'class{i}' for i in range(284)]
    synth.write_dataset(root, n_scans=4, n_frames=5, height=120, width=160, n_boxes=6, class_names=names, seed=2, n_voxels=(8, 8, 4),
                        render_device='cuda')
    cfg = load_config(os.path.join(ROOT, 'configs', 'mv_3ddet.py'))
    pipe = [dict(type='LoadAnnotations3D'), dict(type='MultiViewPipeline', n_images=4, transforms=[
                dict(type='LoadImageFromFile'), dict(type='LoadDepthFromFile'), dict(type='ConvertRGBDToPoints', coord_type='CAMERA'),
                dict(type='PointSample', num_points=2000), dict(type='Resize', scale=(160, 128), keep_ratio=False)]),
            dict(type='AggregateMultiViewPoints', coord_type='DEPTH'), dict(type='PointSample', num_points=6000)]
    ds = EmbodiedScanDataset(root, 'embodiedscan_infos_train.pkl', metainfo=dict(classes=names), pipeline=pipe)
    ld = ScanLoader(ds, batch_size=2, shuffle=False, seed=0, num_threads=2, prefetch=2, pin=True, workers='thread', device_draws=True)
    det = build_detector(cfg, device=dev, seed=0).to(dev)
    optim = build_optim_wrapper(cfg)
    E.PRECISION[0] = 'bf16'
    try:
        n = 0
        for batch in ld:
            assert all('sel_pix' not in s and s['draw'][1:] == (2000, 6000) for s in batch)
            slots = [pipeline.alloc_slot(s, dev) for s in batch]
            dscans = [pipeline.upload_into(sl, s) for sl, s in zip(slots, batch)]      # (+ the draws, on the upload's stream)
            assert all(d['sel_pix'].numel() == 6000 for d in dscans)
            d0 = dscans[0]
            dep = d0['depth'].reshape(d0['depth'].shape[0], -1)
            assert bool((dep[d0['sel_view'].long(), d0['sel_pix'].long()] != 0).all())
            losses = det.train_step(pipeline.make_batch(dscans), optim)
            torch.cuda.synchronize()
            assert all(np.isfinite(float(v)) for v in losses.values())
            n += 1
        assert n == 2
    finally:
        E.PRECISION[0] = 'f32'
    print('train steps fed from files with device-side PointSample draws: ok')
