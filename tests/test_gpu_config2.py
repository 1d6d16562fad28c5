"""Parity AT THE BENCHMARKED CONFIGURATION (BASELINE.json config 2: 20 views of 480x640, 100 k points per scan, the
bench's default schedule: bf16 matrix cores, four HIP streams, tap-split and big-tile kernels live) against the CPU
oracle.  Integer outputs (voxel coordinates of every level, level sizes, target labels and assigned boxes) are checked
bit for bit on the full batch of 4 scans; head logits and the three losses of one scan are checked against the oracle's
f32 forward with the tolerances `north_star` asks to be stated: printed next to the measured error."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
CFG = 'configs/mv_3ddet.py'
MEAN, STD = [123.675, 116.28, 103.53], [58.395, 57.12, 57.375]


def _rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.fixture(scope='module')
def setup():
    import os
    from embodiedscan_amd.config import build_detector
    from embodiedscan_amd.synth import make_scan
    from embodiedscan_amd import pipeline
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    dev = torch.device('cuda:0')
    det = build_detector(os.path.join(root, CFG), device=dev, seed=0).to(dev)
    g = torch.Generator().manual_seed(1)
    sd = {k: v.cpu() for k, v in det.state_dict().items()}
    for k in sd:                    # non-trivial frozen-BN statistics
        if k.startswith('backbone.') and k.endswith('running_var'):
            sd[k] = torch.rand(sd[k].shape, generator=g) + 0.5
        if k.startswith('backbone.') and (k.endswith('running_mean') or k.endswith('bn1.bias') or k.endswith('bn2.bias')):
            sd[k] = torch.randn(sd[k].shape, generator=g) * 0.1
    det.load_state_dict({k: v.to(dev) for k, v in sd.items()})
    scans = [make_scan(1234 + i, render_device='cuda:0') for i in range(4)]       # the bench's scans of rank 0
    dscans = [pipeline.upload_scan(s, dev) for s in scans]
    return det, scans, dscans, sd


def _forward(det, dscans, mode, backward=False, seeds=None):
    from embodiedscan_amd import engine as E, pipeline
    E.PRECISION[0] = mode
    try:
        E.TAPE.clear()
        E.WEIGHT_VERSION[0] += 1
        batch = pipeline.make_batch(dscans)
        points_host = [p.cpu() for p in batch['inputs']['points']]
        data = det.data_preprocessor(batch, True)
        det._bind()
        det.arena.grad.zero_()
        losses = det.forward(data['inputs'], data['data_samples'], mode='loss')
        if seeds is not None:                      # start the backward pass from a given head-output gradient
            for lv, g in zip(det.bbox_head.last_levels, seeds):
                assert lv['ho'].g.shape == g.shape
                lv['ho'].g.copy_(g)
        if backward:
            det._backward(None)
        else:
            E.TAPE.clear()
            E.join_wgrad_streams()
        torch.cuda.synchronize()
    finally:
        E.PRECISION[0] = 'f32'
    return losses, points_host


def _oracle_levels(points_np, n_batch, voxel_size=0.01):
    """integer side of the oracle's forward only (no convolutions): voxel set, backbone level sets, head level sets"""
    from oracle import coords as C
    c, _ = C.voxelize(points_np, voxel_size)
    cur, ts = C.stride_coords(C.stride_coords(c, 2), 4), 4
    lv = []
    for _ in range(4):
        ts *= 2
        cur = C.stride_coords(cur, ts)
        lv.append(cur)
    heads, x = [None] * 4, lv[3]
    heads[3] = x
    for i in (2, 1, 0):
        ch = C.gen_transpose_coords(x, 8 * 2 ** (i + 1))
        x, _, _ = C.union_coords(ch, lv[i], n_batch)            # (round 6 row-order spec: generated children first)
        heads[i] = x
    return c, lv, heads


def _keys(c):
    c = c.astype(np.int64)
    return ((c[:, 0] << 54) | ((c[:, 1] + (1 << 17)) << 36) | ((c[:, 2] + (1 << 17)) << 18) | (c[:, 3] + (1 << 17)))


def test_integer_outputs_batch4_bit_exact(setup):
    """4 scans x 20 views x 100 k points, bf16 default schedule: voxel coordinates of every head level (values AND row
    order), level sizes, target labels, assigned boxes and centerness targets equal the oracle's bit for bit.
    At this size the finest level of some samples exceeds pts_prune_threshold = 100 000 rows, so FCAF3D's pruning is
    live: WHICH rows survive is a float decision (top-k of interpolated scores, checked against the full oracle in the
    one-scan tests below and in test_gpu_model.py::test_prune_path_and_ragged_batch); here the pruned samples must keep
    exactly `thr` rows forming an order-preserving subset of the oracle's candidate set."""
    from oracle import geometry as G
    det, scans, dscans, sd = setup
    losses, points_host = _forward(det, dscans, 'bf16')
    _, lv, heads = _oracle_levels([p.numpy() for p in points_host], 4)
    levels = det.bbox_head.last_levels
    thr = det.bbox_head.pts_prune_threshold
    hips = [levels[l]['cs'].coords.cpu().numpy() for l in range(4)]
    for l in (3, 2, 1):
        np.testing.assert_array_equal(hips[l], heads[l])
    pruned = []
    for b in range(4):
        cand, got = heads[0][heads[0][:, 0] == b], hips[0][hips[0][:, 0] == b]
        if len(cand) <= thr:
            np.testing.assert_array_equal(got, cand)
        else:
            pruned.append(b)
            assert len(got) == thr, (b, len(got), len(cand))
            pos = {int(k): i for i, k in enumerate(_keys(cand))}
            idx = np.array([pos.get(int(k), -1) for k in _keys(got)])
            assert (idx >= 0).all() and (np.diff(idx) > 0).all()           # subset, original row order kept
    sizes = [h.shape[0] for h in hips]
    print(f'head level rows (batch 4, fine->coarse): {sizes}; candidates at level 0: {heads[0].shape[0]}; '
          f'pruned samples: {pruned}; coordinates bit-exact')
    assert sizes[0] > 300000           # the regime the fast kernels / tap split are built for
    tg = det.bbox_head.last_targets
    n_pos = 0
    for b in range(4):
        pts = [torch.from_numpy(h[h[:, 0] == b][:, 1:]).float() * 0.01 for h in hips]
        ct, bt, kt = G.get_targets(pts, torch.from_numpy(scans[b]['gt_boxes']), torch.from_numpy(scans[b]['gt_labels']))
        np.testing.assert_array_equal(tg[b][2].cpu().numpy(), kt.numpy())
        np.testing.assert_array_equal(tg[b][1].cpu().numpy(), bt.numpy())
        np.testing.assert_array_equal(tg[b][0].cpu().numpy(), ct.numpy())
        n_pos += int((kt >= 0).sum())
    print(f'target labels / boxes / centerness bit-exact on {sum(sizes)} locations, {n_pos} positives')
    assert all(np.isfinite(float(v)) for v in losses.values())


@pytest.fixture(scope='module')
def oracle_one_scan(setup):
    from oracle import model as OM
    from embodiedscan_amd import pipeline
    det, scans, dscans, sd = setup
    pts = [pipeline.depth_to_points(dscans[0]).cpu()]
    imgs = OM.preprocess_img(torch.from_numpy(scans[0]['img']), MEAN, STD)[None]
    with torch.no_grad():
        ol, aux = OM.detector_loss(sd, pts, imgs, [scans[0]['meta']], [torch.from_numpy(scans[0]['gt_boxes'])],
                                   [torch.from_numpy(scans[0]['gt_labels'])], return_aux=True, training=True)
    return ol, aux


# stated tolerances at config 2 (relative): losses, head logits (relative L2 per level over centerness + class logits),
# decoded box distances.  f32 = exact-f32 matrix cores; bf16 = the bench default.
# (measured on MI355X: f32 logits 6e-8 / bbox 3e-7 / losses 2e-7; bf16 logits 1.7e-4 / bbox 1.2e-3 / losses 2e-4)
TOL = {'f32': dict(loss=1e-4, logits=1e-5, bbox=1e-5), 'bf16': dict(loss=2e-2, logits=5e-3, bbox=1e-2)}


@pytest.mark.parametrize('mode', ['bf16', 'f32'])
def test_losses_and_logits_one_scan(setup, oracle_one_scan, mode):
    det, scans, dscans, sd = setup
    ol, aux = oracle_one_scan
    losses, _ = _forward(det, dscans[:1], mode)
    tol = TOL[mode]
    levels = det.bbox_head.last_levels
    same_rows = True
    for l in range(4):
        ho = levels[l]['ho'].d.cpu()
        bb = levels[l]['bbox'].cpu()
        oc, ob, ok, opts = aux['outs'][l][0]
        # rows are matched by voxel coordinate: identical sets except where pruning is live (level 0 above 100 k rows),
        # where bf16 rounding of near-tied scores may swap a few rows at the top-k boundary (stated: <= 0.5 % of rows)
        hk = _keys(levels[l]['cs'].coords.cpu().numpy())
        okk = _keys(np.concatenate([np.zeros((opts.shape[0], 1), np.int64),
                                    np.rint(opts.numpy().astype(np.float64) / 0.01).astype(np.int64)], 1))
        if hk.shape == okk.shape and (hk == okk).all():
            ih = io = np.arange(len(hk))
        else:
            same_rows = False
            _, ih, io = np.intersect1d(hk, okk, return_indices=True)
            frac = 1.0 - len(ih) / max(len(okk), 1)
            print(f'{mode} level {l}: {len(hk)} rows vs oracle {len(okk)}, {frac:.3%} of the oracle rows not kept (tol 0.5 %)')
            assert len(hk) == len(okk) and frac <= (5e-3 if mode == 'bf16' else 1e-4)
        ih, io = torch.from_numpy(ih), torch.from_numpy(io)
        e_logit = _rel(torch.cat([ho[:, 0:1], ho[:, 13:13 + ok.shape[1]]], 1)[ih], torch.cat([oc, ok], 1)[io])
        e_box = _rel(bb[ih], ob[io])
        print(f'{mode} level {l} ({ho.shape[0]} rows): logits rel-L2 {e_logit:.2e} (tol {tol["logits"]:.0e}), '
              f'decoded bbox rel-L2 {e_box:.2e} (tol {tol["bbox"]:.0e})')
        assert e_logit < tol['logits'] and e_box < tol['bbox']
    if same_rows:
        np.testing.assert_array_equal(det.bbox_head.last_targets[0][2].cpu().numpy(), aux['targets'][0][2].numpy())
    for k in ol:
        e = abs(float(losses[k]) - float(ol[k])) / abs(float(ol[k]))
        print(f'{mode} {k}: hip {float(losses[k]):.6f} oracle(f32) {float(ol[k]):.6f} rel err {e:.2e} (tol {tol["loss"]:.0e})')
        assert e < tol['loss']


def test_run_to_run_noise_is_bounded(setup):
    """Round 3: NO float atomics are left on the mv-3ddet step (weight gradients: row slices through a workspace reduced in
    slice order; tap-split forward / dgrad: the same; projection-fusion backward: a gather in ascending voxel order; Scale
    gradient: fixed-order block partials; loss values: f64 sums) -- two runs of the benchmarked step (batch 4, bf16, four
    streams) on the same inputs / weights must agree BIT FOR BIT: head logits, losses, pruned sets, targets and every
    parameter gradient.  (Rounds 1-2 bounded the noise of f32 atomics here: logits 5e-4, gradients 1e-6 worst.)
    Second part: exact-f32 and bf16 backward from a fixed head-output gradient, 2 scans: bit-identical as well."""
    det, scans, dscans, sd = setup
    runs = []
    for _ in range(2):
        losses, _ = _forward(det, dscans, 'bf16', backward=True)
        lv = det.bbox_head.last_levels
        runs.append(dict(losses={k: float(v) for k, v in losses.items()}, ho=[l['ho'].d.clone() for l in lv],
                         keys=[_keys(l['cs'].coords.cpu().numpy()) for l in lv],
                         kt=[t[2].clone() for t in det.bbox_head.last_targets],
                         grads={k: v.clone() for k, v in det.arena.grad_dict().items()}))
    a, b = runs
    for l in range(4):
        assert a['keys'][l].shape == b['keys'][l].shape and (a['keys'][l] == b['keys'][l]).all(), f'level {l}: voxel sets differ'
        assert torch.equal(a['ho'][l], b['ho'][l]), f'level {l}: logits differ by {float((a["ho"][l] - b["ho"][l]).abs().max()):.3e}'
    for s in range(4):
        assert torch.equal(a['kt'][s], b['kt'][s])
    assert a['losses'] == b['losses'], (a['losses'], b['losses'])
    diff = {k: float((a['grads'][k] - b['grads'][k]).abs().max()) for k in a['grads']}
    bad = {k: v for k, v in diff.items() if v != 0.0}
    print(f'run-to-run (batch 4, bf16, four streams): logits, losses, targets identical; {len(diff) - len(bad)}/{len(diff)} gradient tensors '
          f'bit-identical' + (f'; differing: {dict(list(bad.items())[:6])}' if bad else ''))
    assert not bad
    two = [dscans[0], dscans[2]]
    for mode in ('f32', 'bf16'):
        _forward(det, two, mode, backward=True)
        seeds = [l['ho'].g.clone() for l in det.bbox_head.last_levels]
        g = []
        for _ in range(2):
            _forward(det, two, mode, backward=True, seeds=seeds)
            g.append({k: v.clone() for k, v in det.arena.grad_dict().items()})
        bad = [k for k in g[0] if not torch.equal(g[0][k], g[1][k])]
        print(f'run-to-run backward, {mode} (same head-output gradient): {len(g[0]) - len(bad)}/{len(g[0])} tensors bit-identical')
        assert not bad, bad[:8]
