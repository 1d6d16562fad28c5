"""Pin the CPU oracle against golden vectors produced by the REFERENCE's own code
(oracle/make_golden.py ran the reference functions; fixtures in tests/golden/)."""
import os
import numpy as np
import torch
from oracle import geometry as G
from oracle import model as M


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + '.npz'))


def test_rotation_and_corners(golden_dir):
    d = _load(golden_dir, 'euler')
    rot = G.rotation_3d_in_euler(torch.from_numpy(d['points']), torch.from_numpy(d['angles']))
    np.testing.assert_allclose(rot.numpy(), d['rotated'], rtol=0, atol=2e-6)
    c = G.euler_box_corners(torch.from_numpy(d['boxes']))
    np.testing.assert_allclose(c.numpy(), d['corners'], rtol=0, atol=2e-6)


def test_unproject_and_aggregate(golden_dir):
    d = _load(golden_dir, 'unproject')
    pts, nz = M.unproject_depth(torch.from_numpy(d['depth']), torch.from_numpy(d['cam2img']))
    np.testing.assert_array_equal(pts[nz].numpy(), d['cam_points'])          # same torch ops -> bit exact
    gp = M.aggregate_points(pts[nz], torch.from_numpy(d['global2cam']))
    np.testing.assert_array_equal(gp.numpy(), d['global_points'])


def test_preprocess_img(golden_dir):
    d = _load(golden_dir, 'preprocess_img')
    out = M.preprocess_img(torch.from_numpy(d['img']), [123.675, 116.28, 103.53], [58.395, 57.12, 57.375])
    np.testing.assert_array_equal(out.numpy(), d['out'])


def _meta_from(d):
    meta = {}
    flow = [str(x) for x in d['flow']]
    if flow:
        meta = dict(pcd_rotation=d['pcd_rotation'], pcd_scale_factor=float(d['pcd_scale_factor']), pcd_trans=d['pcd_trans'],
                    pcd_horizontal_flip=bool(d['hflip']), pcd_vertical_flip=bool(d['vflip']), transformation_3d_flow=flow)
    return meta


def test_batch_point_sample(golden_dir):
    for name in ('point_sample_plain', 'point_sample_aug'):
        d = _load(golden_dir, name)
        out = M.batch_point_sample(_meta_from(d), torch.from_numpy(d['feats']), torch.from_numpy(d['points']),
                                   torch.from_numpy(d['proj']), torch.from_numpy(d['scale_factor']),
                                   torch.from_numpy(d['crop_offset']), bool(d['flip']), tuple(d['pad_shape']),
                                   tuple(d['img_shape']))
        assert (d['out'] != 0).any(1).sum() > 50          # the case really samples something
        np.testing.assert_array_equal(out.numpy(), d['out'])


def test_get_targets(golden_dir):
    for name in ('get_targets', 'get_targets_empty'):
        d = _load(golden_dir, name)
        pts = [torch.from_numpy(d[f'points{i}']) for i in range(4)]
        ct, bt, kt = G.get_targets(pts, torch.from_numpy(d['gt_boxes']), torch.from_numpy(d['gt_labels']))
        np.testing.assert_array_equal(kt.numpy(), d['cls_targets'])             # labels: bit exact
        np.testing.assert_allclose(bt.numpy(), d['bbox_targets'], rtol=0, atol=0)
        pos = d['cls_targets'] >= 0
        np.testing.assert_allclose(ct.numpy()[pos], d['center_targets'][pos], rtol=1e-5, atol=1e-6)


def test_box_coder_and_cd_loss(golden_dir):
    d = _load(golden_dir, 'box_coder_cdloss')
    dec = G.bbox_pred_to_bbox(torch.from_numpy(d['points']), torch.from_numpy(d['pred']))
    np.testing.assert_allclose(dec.numpy(), d['decoded'], rtol=1e-5, atol=2e-6)
    loss = G.bbox_cd_loss(torch.from_numpy(d['decoded']), torch.from_numpy(d['target']))
    np.testing.assert_allclose(loss.numpy(), d['loss'], rtol=1e-6)
