"""N4, device-side PointSample (ScanPipeline(device_draws=True)): the LAW of the counter-based draws and the host-side hand-over.
The kernels are restated integer for integer in oracle/draws.py (GPU equality: tests/test_gpu_draws.py); here the restatement
is held to the law of the reference's `np.random.choice(range(n), k, replace=False)` (datasets/transforms/points.py:189-206):
every element equally likely to be chosen, every chosen element equally likely at every output position, draws of different
seeds / streams independent -- by chi-square tests at a fixed seed set (deterministic: no flaky thresholds)."""
import numpy as np
import pytest


def _chi2_p(obs, exp):
    from scipy import stats
    return float(stats.chisquare(obs, exp).pvalue)


def test_counter_based_draws_follow_the_uniform_law():
    from oracle import draws as D
    n, k, trials = 400, 40, 6000
    idx = np.arange(n, dtype=np.int64)
    incl = np.zeros(n)
    first = np.zeros(n)
    pos_of_0 = np.zeros(k)
    pair = np.zeros((2, 2))
    valid = np.ones(n, bool)
    for t in range(trials):
        sel = D.draw(D.key30(1000 + t, 3, idx), valid, k)
        assert len(np.unique(sel)) == k
        incl[sel] += 1
        first[sel[0]] += 1
        w = np.flatnonzero(sel == 0)
        if len(w):
            pos_of_0[w[0]] += 1
        pair[int(7 in sel), int(8 in sel)] += 1
    # inclusion: every element with probability k / n
    assert _chi2_p(incl, np.full(n, trials * k / n)) > 1e-3
    # the first output position is uniform over the n elements; element 0, when chosen, is uniform over the k positions
    assert _chi2_p(first, np.full(n, trials / n)) > 1e-3
    assert _chi2_p(pos_of_0, np.full(k, pos_of_0.sum() / k)) > 1e-3
    # joint inclusion of two neighbours: hypergeometric (k/n)((k-1)/(n-1)) etc.
    p11 = k / n * (k - 1) / (n - 1)
    p10 = k / n * (n - k) / (n - 1)
    p00 = (n - k) / n * (n - k - 1) / (n - 1)
    assert _chi2_p(pair.reshape(-1), trials * np.array([p00, p10, p10, p11])) > 1e-3
    # different streams of one seed are different draws
    a, b = D.draw(D.key30(5, 0, idx), valid, k), D.draw(D.key30(5, 1, idx), valid, k)
    assert len(set(a) & set(b)) < k // 2
    # invalid elements are never chosen
    valid2 = np.ones(n, bool)
    valid2[::3] = False
    sel = D.draw(D.key30(9, 0, idx), valid2, k)
    assert valid2[sel].all()


def test_point_sample_two_stage_shape_and_validity():
    from oracle import draws as D
    rng = np.random.default_rng(0)
    depth = (rng.random((5, 24, 32)) > 0.3).astype(np.float32) * (1 + rng.random((5, 24, 32)).astype(np.float32))
    sv, sp = D.point_sample(depth, 77, 100, 350)
    assert sv.shape == sp.shape == (350,) and sv.dtype == np.int32
    assert (depth.reshape(5, -1)[sv, sp] != 0).all()
    assert len(set(zip(sv.tolist(), sp.tolist()))) == 350
    assert np.bincount(sv, minlength=5).max() <= 100
    sv2, sp2 = D.point_sample(depth, 78, 100, 350)
    assert not (np.array_equal(sv, sv2) and np.array_equal(sp, sp2))


def test_pipeline_hands_over_a_seed_instead_of_indices(tmp_path):
    """ScanPipeline(device_draws=True): a scan whose frames all hold enough valid pixels carries `draw` = (seed, view_points,
    n_points) and no index arrays; a scan with a nearly empty depth frame falls back to the host's draws"""
    import os
    from embodiedscan_amd import pipeline, synth
    from embodiedscan_amd.datasets import EmbodiedScanDataset
    root = str(tmp_path)
    names = [f'class{i}' for i in range(20)]
    synth.write_dataset(root, n_scans=2, n_frames=4, height=48, width=64, n_boxes=3, class_names=names, seed=3, n_voxels=(8, 8, 4),
                        render_device='cpu')
    pipe = [dict(type='LoadAnnotations3D'), dict(type='MultiViewPipeline', n_images=3, transforms=[
                dict(type='LoadImageFromFile'), dict(type='LoadDepthFromFile'), dict(type='ConvertRGBDToPoints', coord_type='CAMERA'),
                dict(type='PointSample', num_points=200), dict(type='Resize', scale=(64, 48), keep_ratio=False)]),
            dict(type='AggregateMultiViewPoints', coord_type='DEPTH'), dict(type='PointSample', num_points=500)]
    ds = EmbodiedScanDataset(root, 'embodiedscan_infos_train.pkl', metainfo=dict(classes=names), pipeline=pipe)
    ds.pipeline.device_draws = True
    scan = ds.load_scan(0, np.random.RandomState(1))
    assert 'sel_pix' not in scan and scan['draw'][1:] == (200, 500)
    host = pipeline._host_tensors(scan)
    assert 'sel_pix' not in host and 'depth' in host
    assert pipeline._finish({}, scan)['draw'] == scan['draw']
    ds.pipeline.view_points = 10 ** 6                      # more than any frame holds: the reference would draw WITH replacement
    scan = ds.load_scan(0, np.random.RandomState(1))
    assert 'draw' not in scan and len(scan['sel_pix']) == 500
