"""End-to-end parity of the MI355X train-step forward/backward against the CPU oracle on identical
seeded inputs (BASELINE config 1 shape: scans x 4 views of 240x320, random weights)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
CFG = 'configs/mv_3ddet.py'


def _relerr(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


@pytest.fixture(scope='module')
def setup():
    import os
    from embodiedscan_amd.config import build_detector, load_config
    from embodiedscan_amd.synth import make_scan
    from embodiedscan_amd import pipeline
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    dev = torch.device('cuda:0')
    det = build_detector(os.path.join(root, CFG), device=dev, seed=0).to(dev)
    # non-trivial frozen-BN statistics so the folded affine is exercised
    g = torch.Generator().manual_seed(1)
    sd = det.state_dict()
    for k in sd:
        if k.startswith('backbone.') and k.endswith('running_var'):
            sd[k] = torch.rand(sd[k].shape, generator=g) + 0.5
        if k.startswith('backbone.') and (k.endswith('running_mean') or k.endswith('bn1.bias') or k.endswith('bn2.bias')):
            sd[k] = torch.randn(sd[k].shape, generator=g) * 0.1
    det.load_state_dict({k: v.to(dev) for k, v in sd.items()})
    scans = [make_scan(s, n_views=4, height=240, width=320, img_size=(256, 256), n_points=20000) for s in (11, 12)]
    dscans = [pipeline.upload_scan(s, dev) for s in scans]
    return det, scans, dscans, sd


def test_data_side_kernels(setup):
    """A1-A3 (+augmentation) and A18 against the oracle; float tolerance (the oracle solves an LU system where the
    kernel multiplies by the inverse), stated per check."""
    from embodiedscan_amd import pipeline
    from oracle import pipeline as OP, model as OM
    det, scans, dscans, _ = setup
    for s, d in zip(scans, dscans):
        p = pipeline.depth_to_points(d).cpu()
        po = OP.scan_to_points(s)
        err = float((p - po).abs().max())
        print(f'A1-A3 max abs err {err:.3e} m (tol 2e-5)')
        assert err < 2e-5
    data = det.data_preprocessor({'inputs': {'img': torch.stack([d['img'] for d in dscans])}, 'data_samples': None}, True)
    img = data['inputs']['imgs'].cpu()
    ref = torch.stack([OM.preprocess_img(torch.from_numpy(s['img']), [123.675, 116.28, 103.53], [58.395, 57.12, 57.375])
                       for s in scans])
    err = float((img - ref).abs().max())
    print(f'A18 max abs err {err:.3e} (tol 1e-6)')
    assert img.shape == ref.shape and err < 1e-6


def test_train_step_parity(setup):
    from embodiedscan_amd import engine as E, pipeline
    from oracle import model as OM
    det, scans, dscans, sd = setup
    dev = torch.device('cuda:0')
    batch = pipeline.make_batch(dscans)
    points_host = [p.cpu() for p in batch['inputs']['points']]         # identical points for both paths
    E.TAPE.clear()
    data = det.data_preprocessor(batch, True)
    det._bind()
    det.arena.grad.zero_()
    losses = det.forward(data['inputs'], data['data_samples'], mode='loss')
    E.TAPE.backward()
    torch.cuda.synchronize()
    # ---- oracle on the same inputs
    train = set(det.arena.trainable_names())
    ref_names = det.arena.grad_dict().keys()
    osd = {k: v.clone().requires_grad_(k in ref_names) for k, v in sd.items()}
    imgs = torch.stack([OM.preprocess_img(torch.from_numpy(s['img']), [123.675, 116.28, 103.53], [58.395, 57.12, 57.375])
                        for s in scans])
    olosses, aux = OM.detector_loss(osd, points_host, imgs, [s['meta'] for s in scans],
                                    [torch.from_numpy(s['gt_boxes']) for s in scans],
                                    [torch.from_numpy(s['gt_labels']) for s in scans], return_aux=True)
    sum(olosses.values()).backward()
    # integer outputs: level sizes, coordinates and target labels are bit exact
    tg = det.bbox_head.last_targets
    for b in range(len(scans)):
        np.testing.assert_array_equal(tg[b][2].cpu().numpy(), aux['targets'][b][2].numpy())
        np.testing.assert_array_equal(tg[b][1].cpu().numpy(), aux['targets'][b][1].numpy())
    for k in olosses:
        e = abs(float(losses[k]) - float(olosses[k])) / abs(float(olosses[k]))
        print(f'{k}: hip {float(losses[k]):.6f} oracle {float(olosses[k]):.6f} rel err {e:.2e} (tol 1e-3)')
        assert e < 1e-3
    gd = det.arena.grad_dict()
    worst = []
    for k in gd:
        if osd[k].grad is None:
            continue
        worst.append((_relerr(gd[k], osd[k].grad), k))
    worst.sort(reverse=True)
    print('worst gradient rel-to-max errors:', worst[:8], '(tol 2e-2)')
    assert worst[0][0] < 2e-2
    med = float(np.median([w[0] for w in worst]))
    print(f'median gradient rel-to-max error {med:.2e} over {len(worst)} tensors (tol 1e-3)')
    assert med < 1e-3
