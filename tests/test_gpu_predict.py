"""mode='predict' (SURVEY 8f N1): multi-class rotated-BEV NMS kernel and the whole inference path against the oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_nms3d_multiclass_kernel():
    from embodiedscan_amd.hip import P, call
    from oracle import predict as PR
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(5)
    M, C = 300, 6
    centers = torch.rand(40, 3, generator=g) * 4
    boxes = torch.cat([centers[torch.randint(0, 40, (M,), generator=g)] + torch.randn(M, 3, generator=g) * 0.15,
                       torch.rand(M, 3, generator=g) * 1.2 + 0.4, torch.rand(M, 3, generator=g) * 6.2 - 3.1], 1)
    scores = torch.rand(M, C, generator=g)
    scores[:, 5] = 0.0                                  # a class without candidates
    scores[10:20, 0] = scores[10, 0]                    # score ties
    db, ds = boxes.to(dev), scores.to(dev)
    keep_idx = torch.empty((C, M), dtype=torch.int32, device=dev)
    keep_cnt = torch.zeros(C, dtype=torch.int32, device=dev)
    call('es_nms3d_multiclass', P(db), P(ds), M, C, 0.3, 0.25, P(keep_idx), P(keep_cnt), torch.cuda.current_stream().cuda_stream)
    cnt = keep_cnt.cpu().tolist()
    total = 0
    for c in range(C):
        ids = torch.nonzero(scores[:, c] > 0.3).squeeze(1)
        ref = ids[PR.nms3d(boxes[ids][:, :7], scores[ids, c], 0.25)] if ids.numel() else ids
        got = keep_idx[c, :cnt[c]].cpu().long()
        assert got.tolist() == ref.tolist(), (c, got.tolist()[:10], ref.tolist()[:10])
        total += cnt[c]
    assert cnt[5] == 0 and total > 50 and total < M * C


def test_predict_end_to_end():
    import os
    from embodiedscan_amd import engine as E, pipeline
    from embodiedscan_amd.config import build_detector
    from embodiedscan_amd.synth import make_scan
    from oracle import model as OM
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    dev = torch.device('cuda:0')
    det = build_detector(os.path.join(root, 'configs/mv_3ddet.py'), device=dev, seed=0).to(dev)
    # thresholds chosen so that the per-sample top-k selection and the NMS really run on random-init scores
    det.bbox_head.test_cfg = dict(nms_pre=300, iou_thr=0.5, score_thr=0.09)
    g = torch.Generator().manual_seed(2)
    sd = {k: v.cpu() for k, v in det.state_dict().items()}
    for k in sd:                                          # non-trivial running statistics (eval-mode BN is exercised)
        if k.endswith('running_var'):
            sd[k] = torch.rand(sd[k].shape, generator=g) * 0.5 + 0.75
        if k.endswith('running_mean'):
            sd[k] = torch.randn(sd[k].shape, generator=g) * 0.05
    det.load_state_dict({k: v.to(dev) for k, v in sd.items()})
    scans = [make_scan(s, n_views=3, height=120, width=160, img_size=(128, 128), n_points=8000, n_boxes=5) for s in (31, 32)]
    dscans = [pipeline.upload_scan(s, dev) for s in scans]
    batch = pipeline.make_batch(dscans)
    pts_host = [p.cpu() for p in batch['inputs']['points']]
    data = det.data_preprocessor(batch, False)
    out = det.forward(data['inputs'], data['data_samples'], mode='predict')
    torch.cuda.synchronize()
    assert det.training and E.TAPE.enabled                  # predict restores the training state
    imgs = torch.stack([OM.preprocess_img(torch.from_numpy(s['img']), [123.675, 116.28, 103.53], [58.395, 57.12, 57.375])
                        for s in scans])
    ref = OM.detector_predict(sd, pts_host, imgs, [s['meta'] for s in scans], nms_pre=300, score_thr=0.09, iou_thr=0.5)
    n_det = 0
    for ds, (rb, rs, rl) in zip(out, ref):
        pr = ds.pred_instances_3d
        print(f'detections: hip {len(pr.scores_3d)} oracle {len(rs)}')
        assert len(pr.scores_3d) == len(rs)
        np.testing.assert_array_equal(pr.labels_3d.cpu().numpy(), rl.numpy())
        np.testing.assert_allclose(pr.scores_3d.cpu().numpy(), rs.numpy(), rtol=2e-5, atol=1e-7)
        np.testing.assert_allclose(pr.bboxes_3d.tensor.cpu().numpy(), rb.numpy(), rtol=3e-4, atol=2e-4)   # random-init exp() sizes reach 1e2
        n_det += len(rs)
    assert n_det > 20
