"""GPU parity of the occupancy path (BASELINE config 5).

The supervision scatter, the fused CE + sem_scal + geo_scal loss and the dense 3-D neck are compared with the golden
vectors recorded from the REFERENCE's own occ_loss.py / imvoxel_neck.py (tests/golden/occ_*.npz); FPN and the whole
DenseFusionOccPredictor train-step forward/backward are compared with the CPU oracle (oracle/occ.py, itself pinned to the
same golden vectors).  Integer outputs bit exact; float tolerances stated inline."""
import os
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, 'tests', 'golden')
MEAN, STD = [123.675, 116.28, 103.53], [58.395, 57.12, 57.375]


def _rel(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.fixture(scope='module')
def dev():
    return torch.device('cuda:0')


def _rows(t):
    """(1, C, X, Y, Z) -> channels-last rows (X*Y*Z, C)"""
    return t[0].permute(1, 2, 3, 0).reshape(-1, t.shape[1]).contiguous()


def test_occ_targets_and_losses_vs_reference(dev):
    """es_occ_targets / es_occ_loss against the reference's occ_multiscale_supervision, CrossEntropyLoss(ignore 255),
    sem_scal_loss, geo_scal_loss and the autograd gradient of their sum: targets bit exact (incl. duplicate voxels: last
    write wins, masked windows -> 255), values 1e-5, gradient 1e-4 relative L2."""
    from embodiedscan_amd.hip import P, call
    d = np.load(os.path.join(GOLDEN, 'occ_loss.npz'))
    pred = torch.from_numpy(d['pred'])
    C, X, Y, Z = pred.shape[1:]
    occ = torch.from_numpy(d['gt_occ']).to(torch.int32).to(dev)
    mask = torch.from_numpy(d['mask']).to(torch.uint8).to(dev)
    st = torch.cuda.current_stream().cuda_stream
    gts = {}
    for ratio in (1, 2):
        dims = (X // ratio, Y // ratio, Z // ratio)
        nv = dims[0] * dims[1] * dims[2]
        for tag, m in (('_masked', mask), ('', None)):
            gt = torch.empty(nv, dtype=torch.int32, device=dev)
            scratch = torch.empty(nv, dtype=torch.int32, device=dev)
            call('es_occ_targets', P(occ), occ.shape[0], ratio, dims[0], dims[1], dims[2], P(m), P(scratch), P(gt), st)
            np.testing.assert_array_equal(gt.cpu().numpy().reshape(dims), d[f'gt_r{ratio}{tag}'][0])
            gts[(ratio, tag)] = gt
    logits = _rows(pred).to(dev)
    n = logits.shape[0]
    for tag, key in (('masked', (1, '_masked')), ('plain', (1, ''))):
        stats = torch.empty(3 * C + 2, dtype=torch.float64, device=dev)
        coeff = torch.empty(2 * C + 1, dtype=torch.float32, device=dev)
        out = torch.empty(4, dtype=torch.float32, device=dev)
        grad = torch.empty_like(logits)
        call('es_occ_loss', P(logits), C, P(gts[key]), n, C, 1.0, P(stats), P(coeff), P(grad), C, P(out), 0, st)
        torch.cuda.synchronize()
        o = out.cpu().tolist()
        for name, v in zip(('ce', 'sem', 'geo'), o[:3]):
            ref = float(d[f'{name}_{tag}'])
            print(f'occ loss [{tag}] {name}: hip {v:.7f} reference {ref:.7f} (tol 1e-5 rel)')
            assert abs(v - ref) <= 1e-5 * max(1.0, abs(ref))
        e = _rel(grad.cpu(), _rows(torch.from_numpy(d[f'grad_{tag}'])))
        print(f'occ loss [{tag}] gradient rel-L2 {e:.2e} (tol 1e-4)')
        assert e < 1e-4


def _neck_from_golden(dev, precision='f32'):
    from embodiedscan_amd import engine as E
    from embodiedscan_amd.models.necks.imvoxel_neck import IndoorImVoxelNeck
    from embodiedscan_amd.params import ParamArena, imvoxel_neck_specs
    d = np.load(os.path.join(GOLDEN, 'occ_neck.npz'))
    sd = {'neck_3d.' + k[3:]: torch.from_numpy(d[k]) for k in d.files if k.startswith('sd.')}
    arena = ParamArena(imvoxel_neck_specs(in_channels=8, out_channels=4), seed=0)
    missing, unexpected = arena.load_state_dict(sd)
    assert not missing and not unexpected, (missing, unexpected)
    arena.to(dev)
    neck = IndoorImVoxelNeck(8, 4, [1, 1, 1]).bind(arena, 'neck_3d.')
    return d, arena, neck


def test_imvoxel_neck_vs_reference(dev):
    """IndoorImVoxelNeck on the conv engine (dense-grid maps, ConvTranspose3d as generative GEMMs + permutation,
    train-mode BatchNorm3d, residual adds) against the REFERENCE module's forward / input gradient / weight gradients
    (golden).  Exact-f32 kernels; tolerance 2e-4 relative L2 (f32 sums in a different order)."""
    from embodiedscan_amd import engine as E
    d, arena, neck = _neck_from_golden(dev)
    x = torch.from_numpy(d['x'])
    xv = E.Var(_rows(x).to(dev))
    E.TAPE.clear()
    outs = neck(xv, tuple(x.shape[2:]), 1)
    for (o, dims), i in zip(outs, range(3)):
        ref = torch.from_numpy(d[f'out{i}'])
        assert dims == tuple(ref.shape[2:])
        e = _rel(o.d.cpu(), _rows(ref))
        print(f'neck out{i} {tuple(ref.shape)}: rel-L2 {e:.2e} (tol 2e-4)')
        assert e < 2e-4
        o.g = (2 * o.d).contiguous()                       # d/dx of sum(o*o)
    E.TAPE.backward()
    torch.cuda.synchronize()
    e = _rel(xv.g.cpu(), _rows(torch.from_numpy(d['dx'])))
    gd = arena.grad_dict()
    e1 = _rel(gd['neck_3d.down_layer_0.0.conv1.weight'], d['dw_conv1'])
    e2 = _rel(gd['neck_3d.up_block_1.0.weight'], d['dw_up'])
    print(f'neck dx rel-L2 {e:.2e}, dW conv1 {e1:.2e}, dW ConvTranspose3d {e2:.2e} (tol 1e-3)')
    assert max(e, e1, e2) < 1e-3


def test_fpn_vs_oracle(dev):
    """mmdet.FPN (laterals with bias, nearest top-down adds incl. a non-2x size, 3x3 output convs) forward + backward
    against the oracle's F.conv2d / F.interpolate graph.  f32 kernels, tolerance 1e-4."""
    from embodiedscan_amd import engine as E
    from embodiedscan_amd.models.necks.fpn import FPN
    from embodiedscan_amd.params import ParamArena, fpn_specs
    from oracle import occ as OO
    chans, sizes, n_img = (16, 32, 64, 128), ((12, 16), (6, 8), (3, 4), (2, 2)), 3
    arena = ParamArena(fpn_specs(in_channels=chans, out_channels=32), seed=3)
    g = torch.Generator().manual_seed(0)
    for k in arena.p:
        if k.endswith('.bias'):
            arena.p[k].copy_(torch.randn(arena.p[k].shape, generator=g) * 0.1)
    sd = {k: v.clone().requires_grad_(True) for k, v in arena.state_dict().items()}
    arena.to(dev)
    fpn = FPN(list(chans), 32, 4).bind(arena, 'neck.')
    xs = [torch.randn(n_img, c, h, w, generator=g) for c, (h, w) in zip(chans, sizes)]
    feats = [(E.Var(x.permute(0, 2, 3, 1).reshape(-1, x.shape[1]).contiguous().to(dev)), h, w) for x, (h, w) in zip(xs, sizes)]
    E.TAPE.clear()
    outs = fpn(feats, n_img)
    xo = [x.clone().requires_grad_(True) for x in xs]
    oo = OO.fpn(xo, sd)
    tot = 0
    for (o, h, w), r in zip(outs, oo):
        rr = r.permute(0, 2, 3, 1).reshape(-1, r.shape[1])
        e = _rel(o.d.cpu(), rr.detach())
        print(f'FPN out {h}x{w}: rel-L2 {e:.2e} (tol 1e-4)')
        assert e < 1e-4
        o.g = torch.ones_like(o.d)
        tot = tot + r.sum()
    tot.backward()
    E.TAPE.backward()
    torch.cuda.synchronize()
    for (v, h, w), x in zip(feats, xo):
        e = _rel(v.g.cpu(), x.grad.permute(0, 2, 3, 1).reshape(-1, x.shape[1]))
        assert e < 1e-4, e
    gd = arena.grad_dict()
    worst = max(_rel(gd[k], sd[k].grad) for k in gd)
    print(f'FPN parameter gradients worst rel-L2 {worst:.2e} (tol 1e-4)')
    assert worst < 1e-4


def _small_cfg(fpn_out=32, base=16, n_voxels=(8, 8, 4)):
    from embodiedscan_amd.config import load_config
    cfg = load_config(os.path.join(ROOT, 'configs', 'mv_occ.py'))
    m = cfg['model']
    m['backbone']['base_channels'] = base
    m['neck'].update(in_channels=[4 * base, 8 * base, 16 * base, 32 * base], out_channels=fpn_out)
    m['neck_3d']['in_channels'] = fpn_out + 512
    m['n_voxels'] = list(n_voxels)
    return cfg


def _occ_case(dev, cfg, seed=21, views=3):
    from embodiedscan_amd import pipeline
    from embodiedscan_amd.config import build_detector
    from embodiedscan_amd.synth import make_occ_gt, make_scan
    det = build_detector(cfg, device=dev, seed=0).to(dev)
    scan = make_scan(seed, n_views=views, height=120, width=160, img_size=(128, 128), n_points=20000, n_boxes=10, augment=False)
    occ = make_occ_gt(scan, n_voxels=cfg['model']['n_voxels'], prior_range=cfg['prior_generator']['ranges'][0], seed=seed)
    dscan = pipeline.upload_scan(scan, dev)
    return det, scan, occ, dscan


def _oracle_loss(cfg, sd, scan, occ, points_host, grads=None):
    from oracle import model as OM, occ as OO
    imgs = OM.preprocess_img(torch.from_numpy(scan['img']), MEAN, STD)[None]
    m = cfg['model']
    return OO.detector_loss(sd, points_host, imgs, [scan['meta']], [torch.from_numpy(occ['gt_occupancy'])],
                            [torch.from_numpy(occ['gt_occupancy_masks'])], m['n_voxels'], m['point_cloud_range'],
                            cfg['prior_generator']['ranges'][0], tuple(m['neck_3d']['n_blocks']), return_aux=True)


def test_occ_detector_train_step_vs_oracle(dev):
    """DenseFusionOccPredictor forward + backward (narrow 2-D branch, true 3-D widths 544 -> 1088 -> 2176) against the
    oracle: voxelisation / supervision targets bit exact; f32: losses 1e-4, logits 1e-4, parameter gradients median
    1e-3 / worst 5e-2 relative L2; bf16: losses 2e-2, logits 8e-2 (the coarsest level has 4 voxels: train-mode BatchNorm over
    4 rows after 2176-wide bf16 reductions; measured 1.3e-2 / 2.9e-2 / 4.2e-2 fine -> coarse)."""
    from embodiedscan_amd import engine as E, pipeline
    cfg = _small_cfg()
    det, scan, occ, dscan = _occ_case(dev, cfg)
    sd = {k: v.cpu() for k, v in det.state_dict().items()}
    names = set(det.arena.grad_dict().keys())
    osd = {k: v.clone().requires_grad_(k in names) for k, v in sd.items()}
    res = {}
    try:
        for mode in ('f32', 'bf16'):
            E.PRECISION[0] = mode
            E.WEIGHT_VERSION[0] += 1
            E.TAPE.clear()
            batch = pipeline.make_occ_batch([dscan], [occ])
            points_host = [p.cpu() for p in batch['inputs']['points']]
            data = det.data_preprocessor(batch, True)
            det._bind()
            det.arena.grad.zero_()
            losses = det.forward(data['inputs'], data['data_samples'], mode='loss')
            E.TAPE.backward()
            torch.cuda.synchronize()
            res[mode] = dict(losses={k: float(v) for k, v in losses.items()},
                             logits=[l['logits'].d.cpu() for l in det.bbox_head.last],
                             gt=[l['gt'].cpu() for l in det.bbox_head.last],
                             grads={k: v.cpu() for k, v in det.arena.grad_dict().items()})
    finally:
        E.PRECISION[0] = 'f32'
    ol, aux = _oracle_loss(cfg, osd, scan, occ, points_host)
    sum(ol.values()).backward()
    for i in range(3):
        np.testing.assert_array_equal(res['f32']['gt'][i].numpy(), aux['parts'][i][3].reshape(-1).numpy())
    for mode, tl, tg in (('f32', 1e-4, 1e-4), ('bf16', 2e-2, 8e-2)):
        for i in range(3):
            e = _rel(res[mode]['logits'][i], _rows(aux['preds'][i].detach()))
            print(f'{mode} occ logits level {i}: rel-L2 {e:.2e} (tol {tg:.0e})')
            assert e < tg
        for k in ol:
            e = abs(res[mode]['losses'][k] - float(ol[k])) / abs(float(ol[k]))
            print(f'{mode} {k}: hip {res[mode]["losses"][k]:.6f} oracle {float(ol[k]):.6f} rel err {e:.2e} (tol {tl:.0e})')
            assert e < tl
    rel = {k: _rel(v, osd[k].grad) for k, v in res['f32']['grads'].items() if osd[k].grad is not None and float(osd[k].grad.norm()) > 1e-10}
    worst = max(rel, key=rel.get)
    med = float(np.median(list(rel.values())))
    print(f'f32 parameter gradients vs oracle autograd: {len(rel)} tensors, median rel-L2 {med:.2e} (tol 1e-3), worst {rel[worst]:.2e} at {worst} (tol 5e-2)')
    assert med < 1e-3 and rel[worst] < 5e-2
    assert torch.isfinite(det.arena.grad).all()
    # bf16 mode against its own arithmetic specification (oracle/rounding.py: products on bf16-rounded operands, f32
    # accumulation): logits 5e-2 (measured 8e-3 / 2.2e-2 / coarsest level: train-mode BatchNorm over 4 .. 256 rows re-normalises summation-order noise), losses 5e-3, parameter gradients median 5e-3, 90 % of the tensors 3e-2, worst 3e-1 -- a bf16
    # dgrad / wgrad that is wrong on one tensor fails this, which the bf16-vs-f32 comparison above cannot see
    from oracle import rounding as R
    osd2 = {k: v.clone().requires_grad_(k in names) for k, v in sd.items()}
    with R.bf16_operands(act16=True):              # (round 5: the occupancy detector's image backbone stores bf16 activations too)
        ol2, aux2 = _oracle_loss(cfg, osd2, scan, occ, points_host)
        sum(ol2.values()).backward()
    for i in range(3):
        e = _rel(res['bf16']['logits'][i], _rows(aux2['preds'][i].detach()))
        print(f'bf16 occ logits level {i} vs bf16-operand oracle: rel-L2 {e:.2e} (tol 5e-2)')
        assert e < 5e-2
    for k in ol2:
        e = abs(res['bf16']['losses'][k] - float(ol2[k])) / abs(float(ol2[k]))
        print(f'bf16 {k} vs bf16-operand oracle: rel err {e:.2e} (tol 5e-3)')
        assert e < 5e-3
    rel = {k: _rel(v, osd2[k].grad) for k, v in res['bf16']['grads'].items() if osd2[k].grad is not None and float(osd2[k].grad.norm()) > 1e-10}
    v = np.sort(np.array(list(rel.values())))
    worst = max(rel, key=rel.get)
    med, p90 = float(np.median(v)), float(v[int(0.9 * (len(v) - 1))])
    print(f'bf16 parameter gradients vs bf16-operand oracle: {len(v)} tensors, median {med:.2e}, 90th percentile {p90:.2e}, worst '
          f'{rel[worst]:.2e} at {worst} -- loosely bounded (worst < 0.9, median < 0.6), not a tight gate: two bf16 summation orders drift apart by the quantisation noise '
          f'within a few layers and the train-mode BatchNorm backwards over 4 .. 256 rows amplify it; the arithmetic gate of every bf16 '
          f'backward launch (2e-4 on the operands it saw) is tests/test_gpu_insitu.py::test_every_conv_backward_of_a_bf16_occupancy_step_matches_its_specification')
    assert np.isfinite(v).all() and all(bool(torch.isfinite(g).all()) for g in res['bf16']['grads'].values())
    # a LOOSE bound stays: a zeroed tensor scores 1.0, an uncorrelated one (wrong tap mirror / parity class) ~1.41, a
    # flipped sign 2.0; summation-order chaos measures 0.28 - 0.38 median, 0.43 - 0.54 worst (profiles/r5_*)
    assert rel[worst] < 0.9 and med < 0.6, (worst, rel[worst], med)


def test_occ_full_width_forward_and_predict(dev):
    """the shipped widths (ResNet-50 base 64, FPN 256, neck 768 -> 1536 -> 3072: the fast bf16 kernels' shapes) on a small
    8x8x4 volume: bf16 losses within 2e-2 of the f32 oracle, and mode='predict' returns the oracle's arg-max occupancy
    on >= 95 % of the voxels (random-init weights put the 81 class logits of a voxel within a few percent of each other, so
    bf16-sized logit differences flip some arg-maxes; measured 98.4 %)."""
    from embodiedscan_amd import engine as E, pipeline
    from oracle import occ as OO, model as OM
    cfg = _small_cfg(fpn_out=256, base=64)
    det, scan, occ, dscan = _occ_case(dev, cfg, seed=22, views=2)
    sd = {k: v.cpu() for k, v in det.state_dict().items()}
    E.PRECISION[0] = 'bf16'
    try:
        E.TAPE.clear()
        batch = pipeline.make_occ_batch([dscan], [occ])
        points_host = [p.cpu() for p in batch['inputs']['points']]
        data = det.data_preprocessor(batch, True)
        det._bind()
        losses = det.forward(data['inputs'], data['data_samples'], mode='loss')
        E.TAPE.clear()
        E.join_wgrad_streams()
        with torch.no_grad():
            ol, aux = _oracle_loss(cfg, sd, scan, occ, points_host)
        for k in ol:
            e = abs(float(losses[k]) - float(ol[k])) / abs(float(ol[k]))
            print(f'full-width bf16 {k}: hip {float(losses[k]):.6f} oracle {float(ol[k]):.6f} rel err {e:.2e} (tol 2e-2)')
            assert e < 2e-2
        # predict: eval-mode BN (running statistics were just updated by the training forward on both sides? no: the
        # oracle's functional BN does not write back) -> compare against the oracle run in eval mode with HIP's statistics
        sd_eval = {k: v.cpu() for k, v in det.state_dict().items()}
        out = det.forward(data['inputs'], data['data_samples'], mode='predict')
        pred = out[0].pred_occupancy.cpu()
        imgs = OM.preprocess_img(torch.from_numpy(scan['img']), MEAN, STD)[None]
        m = cfg['model']
        with torch.no_grad():
            ref = OO.detector_forward(sd_eval, points_host, imgs, [scan['meta']], m['n_voxels'], m['point_cloud_range'],
                                      cfg['prior_generator']['ranges'][0], training=False)[0]
        want = torch.max(torch.softmax(ref, dim=1), dim=1)[1][0]
        agree = float((pred == want).float().mean())
        print(f'predict: arg-max occupancy agrees on {agree:.2%} of {want.numel()} voxels (tol 95 %)')
        assert pred.shape == want.shape and agree >= 0.95
    finally:
        E.PRECISION[0] = 'f32'
