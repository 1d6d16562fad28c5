"""GPU parity of the grounding path (BASELINE config 4) and the device-side assignment (8f row N2).

Kernel-level: attention forward/backward (exact-f32 and bf16 matrix cores) against torch's scaled-dot-product attention,
LayerNorm / contrastive logits / box coder against torch, the exact oriented-box IoU against the oracle (itself pinned to
qhull), and the whole matching + loss block (costs, Hungarian assignment, labels, focal loss, decoupled corner-Chamfer
loss, gradients) against golden vectors recorded from the REFERENCE's own HungarianAssigner3D / match costs /
GroundingHead.loss_by_feat_single (tests/golden/ground_head.npz).  Model-level: decoder + head and the full grounder
train step against the CPU oracle (oracle/grounding.py).  Assignments bit exact; float tolerances stated inline."""
import os
import numpy as np
import pytest
import torch
import torch.nn.functional as F

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


@pytest.mark.parametrize('bf16,tol', [(0, 2e-5), (1, 2e-2)])
def test_attention_fwd_bwd_vs_torch(dev, bf16, tol):
    """es_attn_fwd / es_attn_bwd (head_dim 32, 8 heads, ragged key lengths, row counts that are not tile multiples,
    operands given as column slices of wider buffers) against softmax(q k^T / sqrt(32)) v and its autograd gradients"""
    from embodiedscan_amd.hip import P, call
    B, H, Lq, Lk, D = 3, 8, 70, 150, 32
    E = H * D
    g = torch.Generator().manual_seed(1)
    qkv = torch.randn(B * Lq, 3 * E, generator=g)            # q lives in a packed buffer (ld = 3E)
    k, v = torch.randn(B * Lk, E, generator=g), torch.randn(B * Lk, E, generator=g)
    do = torch.randn(B * Lq, E, generator=g)
    klen = torch.tensor([150, 37, 97], dtype=torch.int32)
    q = qkv[:, E:2 * E]
    qt, kt, vt = (t.clone().requires_grad_(True) for t in (q, k, v))
    qh = qt.view(B, Lq, H, D).transpose(1, 2)
    kh, vh = kt.view(B, Lk, H, D).transpose(1, 2), vt.view(B, Lk, H, D).transpose(1, 2)
    mask = (torch.arange(Lk)[None, :] < klen[:, None])[:, None, None, :]
    ref = F.scaled_dot_product_attention(qh, kh, vh, attn_mask=mask).transpose(1, 2).reshape(B * Lq, E)
    (ref * do).sum().backward()
    st = torch.cuda.current_stream().cuda_stream
    dq_buf = qkv.to(dev)
    kd, vd, dod, kl = k.to(dev), v.to(dev), do.to(dev), klen.to(dev)
    o = torch.empty(B * Lq, E, device=dev)
    lse = torch.empty(B * H * Lq, device=dev)
    qptr = dq_buf.data_ptr() + 4 * E
    call('es_attn_fwd', qptr, 3 * E, P(kd), E, P(vd), E, B, H, Lq, Lk, P(kl), P(o), E, P(lse), bf16, st)
    dq, dk, dv = torch.empty(B * Lq, E, device=dev), torch.empty_like(kd), torch.empty_like(vd)
    delta = torch.empty(B * H * Lq, device=dev)
    call('es_attn_bwd', qptr, 3 * E, P(kd), E, P(vd), E, P(o), E, P(dod), E, P(lse), B, H, Lq, Lk, P(kl), P(delta), P(dq), E, P(dk), E,
         P(dv), E, 0, bf16, st)
    torch.cuda.synchronize()
    errs = dict(o=_rel(o, ref.detach()), dq=_rel(dq, qt.grad), dk=_rel(dk, kt.grad), dv=_rel(dv, vt.grad))
    print(f'attention bf16={bf16}: ' + '  '.join(f'{n} {e:.2e}' for n, e in errs.items()) + f'  (tol {tol:.0e} rel-L2)')
    assert max(errs.values()) < tol, errs
    # padded keys receive exact zeros
    for b in range(B):
        assert float(dk.view(B, Lk, E)[b, int(klen[b]):].abs().max() if klen[b] < Lk else 0) == 0.0
    # accumulate=1 adds on top
    call('es_attn_bwd', qptr, 3 * E, P(kd), E, P(vd), E, P(o), E, P(dod), E, P(lse), B, H, Lq, Lk, P(kl), P(delta), P(dq), E, P(dk), E,
         P(dv), E, 1, bf16, st)
    torch.cuda.synchronize()
    assert _rel(dq, 2 * qt.grad) < tol and _rel(dv, 2 * vt.grad) < tol


def test_layernorm_contrastive_decode_topk_vs_torch(dev):
    from embodiedscan_amd import hip
    from embodiedscan_amd.hip import P, call
    g = torch.Generator().manual_seed(2)
    st = torch.cuda.current_stream().cuda_stream
    n, C = 333, 256
    x, r = torch.randn(n, C, generator=g), torch.randn(n, C, generator=g)
    w, b, dy = torch.rand(C, generator=g) + .5, torch.randn(C, generator=g), torch.randn(n, C, generator=g)
    xt, rt, wt, bt = (t.clone().requires_grad_(True) for t in (x, r, w, b))
    ref = F.layer_norm(xt + rt, (C,), wt, bt, 1e-5)
    (ref * dy).sum().backward()
    xd, rd, wd, bd, dyd = (t.to(dev) for t in (x, r, w, b, dy))
    y, z = torch.empty(n, C, device=dev), torch.empty(n, C, device=dev)
    mean, rstd = torch.empty(n, device=dev), torch.empty(n, device=dev)
    call('es_layernorm_fwd', P(xd), P(rd), n, C, P(wd), P(bd), 1e-5, P(y), P(z), P(mean), P(rstd), st)
    dz, dw, db = torch.empty(n, C, device=dev), torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    ws = torch.zeros(int(hip.raw('es_layernorm_bwd_workspace_floats')(n, C)), device=dev)
    call('es_layernorm_bwd', P(dyd), P(z), n, C, P(wd), P(mean), P(rstd), P(dz), 0, P(dw), P(db), P(ws), ws.numel(), st)
    dw2, db2 = torch.zeros(C, device=dev), torch.zeros(C, device=dev)          # second launch on the same workspace: the ticket was reset,
    call('es_layernorm_bwd', P(dyd), P(z), n, C, P(wd), P(mean), P(rstd), P(dz), 0, P(dw2), P(db2), P(ws), ws.numel(), st)   # bit-identical sums
    torch.cuda.synchronize()
    assert torch.equal(dw, dw2) and torch.equal(db, db2) and int(ws[:1].view(torch.int32)) == 0
    e = max(_rel(y, ref.detach()), _rel(dz, xt.grad), _rel(dw, wt.grad), _rel(db, bt.grad))
    print(f'LayerNorm(+residual) fwd/bwd worst rel-L2 {e:.2e} (tol 1e-5)')
    assert e < 1e-5 and torch.equal(xt.grad, rt.grad)
    # contrastive logits (+ row max) and their gradients
    B, L, T, Tout = 2, 45, 9, 9
    tl, vl = torch.tensor([9, 6], dtype=torch.int32), torch.tensor([45, 30], dtype=torch.int32)
    v, t = torch.randn(B, L, C, generator=g), torch.randn(B, T, C, generator=g)
    bias = torch.tensor([-4.6])
    vt_, tt_, bt_ = (u.clone().requires_grad_(True) for u in (v, t, bias))
    tm, vm = torch.arange(T)[None] < tl[:, None], torch.arange(L)[None] < vl[:, None]
    ref = vt_ @ tt_.transpose(1, 2) / 16.0 + bt_
    keep = tm[:, None, :] & vm[:, :, None]
    dl = torch.randn(B, L, Tout, generator=g) * keep
    (ref * dl).sum().backward()
    vd, td, bd2, dld = v.to(dev), t.to(dev), bias.to(dev), dl.to(dev).contiguous()
    tld, vld = tl.to(dev), vl.to(dev)                       # keep the device copies alive across the launches
    lo, rm = torch.empty(B, L, Tout, device=dev), torch.empty(B, L, device=dev)
    call('es_contrastive_fwd', P(vd), B, L, P(td), T, C, P(tld), P(vld), P(bd2), P(lo), Tout, P(rm), st)
    dvv, dtt, dbb = torch.empty_like(vd), torch.zeros_like(td), torch.zeros(1, device=dev)
    cws = torch.zeros(int(hip.raw('es_contrastive_bwd_workspace_floats')(B, T)), device=dev)
    call('es_contrastive_bwd', P(dld), Tout, P(vd), B, L, P(td), T, C, P(tld), P(dvv), 0, P(dtt), P(dbb), P(cws), cws.numel(), st)
    dtt2, dbb2 = torch.zeros_like(td), torch.zeros(1, device=dev)
    call('es_contrastive_bwd', P(dld), Tout, P(vd), B, L, P(td), T, C, P(tld), 0, 0, P(dtt2), P(dbb2), P(cws), cws.numel(), st)
    torch.cuda.synchronize()
    assert torch.equal(dtt, dtt2) and torch.equal(dbb, dbb2)                   # deterministic, with and without the dv workgroups
    refm = ref.detach().masked_fill(~keep, float('-inf'))
    assert torch.equal(torch.isinf(lo.cpu()), torch.isinf(refm))
    e = max(_rel(torch.nan_to_num(lo.cpu(), 0, 0, 0), torch.nan_to_num(refm, 0, 0, 0)), _rel(dvv, vt_.grad), _rel(dtt, tt_.grad),
            _rel(dbb, bt_.grad))
    e_max = _rel(torch.nan_to_num(rm.cpu(), 0, 0, 0), torch.nan_to_num(refm.max(-1)[0], 0, 0, 0))
    print(f'ContrastiveEmbed logits / gradients worst rel-L2 {e:.2e}, row max {e_max:.2e} (tol 1e-5)')
    assert e < 1e-5 and e_max < 1e-6
    # sorted top-k (descending, ties -> lower row), valid prefix only
    vals = torch.randn(3, 500, generator=g)
    vals[0, 10] = vals[0, 400] = 7.0
    vlen = torch.tensor([500, 123, 77], dtype=torch.int32)
    idx = torch.empty(3, 64, dtype=torch.int32, device=dev)
    valsd, vlend = vals.to(dev), vlen.to(dev)
    call('es_topk_sorted', P(valsd), 3, 500, P(vlend), 64, P(idx), st)
    for bb in range(3):
        want = torch.argsort(vals[bb, :int(vlen[bb])], descending=True, stable=True)[:64]
        assert torch.equal(idx[bb].cpu().long(), want)
    # box coder
    pred, pts = torch.randn(50, 9, generator=g), torch.randn(50, 3, generator=g)
    pred[:5, 3:6] = -6.0                                     # exp below the 2e-2 clamp
    pt = pred.clone().requires_grad_(True)
    refb = torch.cat((pt[:, :3] + pts, torch.exp(pt[:, 3:6]).clamp(min=2e-2), pt[:, 6:]), -1)
    gb = torch.randn(50, 9, generator=g)
    (refb * gb).sum().backward()
    box, dp = torch.empty(50, 9, device=dev), torch.empty(50, 9, device=dev)
    predd, ptsd, gbd = pred.to(dev), pts.to(dev), gb.to(dev)
    call('es_ground_decode_fwd', P(predd), 9, P(ptsd), 50, P(box), st)
    call('es_ground_decode_bwd', P(predd), 9, P(gbd), 50, P(dp), 9, 0, st)
    torch.cuda.synchronize()
    assert _rel(box, refb.detach()) < 1e-6 and _rel(dp, pt.grad) < 1e-6


def test_fcaf_box_coder_vs_reference(dev):
    """es_ground_decode_fcaf_fwd / _bwd against what the REFERENCE's GroundingHead._bbox_pred_to_bbox(box_coder='FCAF') and its
    autograd produced (tests/golden/ground_coder_fcaf.npz, oracle/make_golden_ground.py): boxes 1e-6, gradient w.r.t. the raw
    regression output 1e-5 (incl. rows whose exp falls below the 2e-2 clamp: zero gradient there); accumulate mode adds"""
    from embodiedscan_amd.hip import P, call
    d = np.load(os.path.join(GOLDEN, 'ground_coder_fcaf.npz'))
    st = torch.cuda.current_stream().cuda_stream
    reg, pts, gb = (torch.from_numpy(d[k]).reshape(-1, d[k].shape[-1]).to(dev).contiguous() for k in ('reg', 'points', 'dboxes'))
    n = reg.shape[0]
    wide = torch.zeros(n, 16, device=dev)                     # the regression output as a column slice of a wider buffer
    wide[:, 3:12] = reg
    box, dp = torch.empty(n, 9, device=dev), torch.full((n, 9), 7.0, device=dev)
    call('es_ground_decode_fcaf_fwd', wide.data_ptr() + 12, 16, P(pts), n, P(box), st)
    call('es_ground_decode_fcaf_bwd', wide.data_ptr() + 12, 16, P(gb), n, P(dp), 9, 0, st)
    torch.cuda.synchronize()
    e1, e2 = _rel(box, d['boxes'].reshape(-1, 9)), _rel(dp, d['dreg'].reshape(-1, 9))
    print(f"FCAF box coder vs the reference: boxes rel-L2 {e1:.2e} (tol 1e-6), d/d reg {e2:.2e} (tol 1e-5); "
          f"{int((torch.exp(reg[:, :6]) < 2e-2).sum())} clamped distances")
    assert e1 < 1e-6 and e2 < 1e-5
    clamped = (torch.exp(reg[:, :6]) < 2e-2)
    assert clamped.any() and float(dp[:, :6][clamped].abs().max()) == 0.0
    call('es_ground_decode_fcaf_bwd', wide.data_ptr() + 12, 16, P(gb), n, P(dp), 9, 1, st)
    torch.cuda.synchronize()
    assert _rel(dp, 2 * d['dreg'].reshape(-1, 9)) < 1e-5


def test_box3d_iou_vs_oracle(dev):
    """es_box3d_iou (N2) against the oracle's exact polyhedral IoU: random oriented pairs + identical / touching /
    contained / disjoint boxes.  f64 geometry; tolerance 1e-6 absolute (f32 output)."""
    from embodiedscan_amd.hip import P, call
    from oracle import grounding as OG
    rng = np.random.default_rng(3)
    a = np.concatenate([rng.uniform(-1, 1, (40, 3)), rng.uniform(.2, 2., (40, 3)), rng.uniform(-3.1, 3.1, (40, 3))], 1).astype(np.float32)
    b = a[rng.integers(0, 40, 9)] + np.concatenate([rng.normal(0, .3, (9, 3)), rng.normal(0, .1, (9, 3)), rng.normal(0, .4, (9, 3))], 1).astype(np.float32)
    b[:, 3:6] = np.abs(b[:, 3:6]) + .05
    b[0] = a[0]                                             # identical
    a[1] = [0, 0, 0, 2, 2, 2, 0, 0, 0]; b[1] = [2, 0, 0, 2, 2, 2, 0, 0, 0]           # touching faces
    b[2] = [50, 50, 50, 1, 1, 1, 0, 0, 0]                   # far away
    out = torch.empty(40, 9, device=dev)
    ad, bd = torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)
    call('es_box3d_iou', P(ad), 40, P(bd), 9, P(out), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    want = OG.overlaps(torch.from_numpy(a), torch.from_numpy(b))
    err = float((out.cpu() - want).abs().max())
    print(f'box3d IoU 40x9 pairs: max abs err {err:.2e} (tol 1e-6); identical {float(out[0, 0]):.6f}, touching {float(out[1, 1]):.6f}')
    assert err < 1e-6 and abs(float(out[0, 0]) - 1.0) < 1e-6 and float(out[1, 1]) == 0.0 and float(out[5, 2]) == 0.0
    assert float((want > 0.05).float().mean()) > 0.03       # the case set does contain real overlaps


def test_matching_and_losses_vs_reference(dev):
    """costs + Hungarian assignment + labels + focal loss + decoupled corner-Chamfer loss, values and gradients, against
    what the REFERENCE's HungarianAssigner3D (scipy), BinaryFocalLossCost / BBox3DL1Cost / IoU3DCost and
    GroundingHead.loss_by_feat_single produced on the same inputs (golden).  Assignment bit exact, cost 1e-4, losses
    1e-5, gradients 1e-4."""
    from embodiedscan_amd.hip import P, call, farr
    from embodiedscan_amd.models.task_modules.assigners import HungarianAssigner3D
    d = np.load(os.path.join(GOLDEN, 'ground_head.npz'))
    B, Q = d['cls'].shape[:2]
    mask = torch.from_numpy(d['mask'])
    T = mask.shape[1]
    tlen = mask.sum(1).to(torch.int32).to(dev)
    logits = torch.from_numpy(d['cls'][:, :, :T]).contiguous()
    logits = torch.where(torch.isinf(logits), torch.zeros_like(logits), logits).to(dev)      # padded tokens are never read
    boxes = torch.from_numpy(d['boxes']).to(dev).contiguous()
    gtb = [torch.from_numpy(d[f'gt_boxes{b}']) for b in range(B)]
    pms = [torch.from_numpy(d[f'pos_map{b}'][:, :T]) for b in range(B)]
    Gs = [len(x) for x in gtb]
    gt_boxes = torch.cat(gtb).to(dev).contiguous()
    pos_map = (torch.cat(pms) != 0).to(torch.uint8).to(dev).contiguous()
    gt_off = torch.tensor([0] + list(np.cumsum(Gs)), dtype=torch.int32, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    asg = HungarianAssigner3D([dict(type='BinaryFocalLossCost', weight=1.0), dict(type='BBox3DL1Cost', weight=2.0),
                               dict(type='IoU3DCost', weight=2.0)])
    q2g = asg.match(logits.view(B, Q, T), boxes.view(B, Q, 9), gt_boxes, pos_map, gt_off, max(Gs), tlen, st)
    torch.cuda.synchronize()
    for b in range(B):
        want = torch.from_numpy(d[f'gt_inds{b}'])
        np.testing.assert_array_equal((q2g[b].cpu() + 1).numpy(), want.numpy())
        cost = asg.last_cost[b, :Gs[b]].cpu().t()                       # (Q, G)
        e = float((cost - torch.from_numpy(d[f'costs{b}'].sum(0)).double()).abs().max())
        print(f'sample {b}: assignment identical to scipy on the reference costs; cost matrix max abs err {e:.2e} (tol 1e-4)')
        assert e < 1e-4
    n_pos = sum(Gs)
    avg = torch.tensor([float(max(n_pos, 1))], device=dev)
    dlog = torch.empty_like(logits)
    lsum = torch.zeros(1, dtype=torch.float64, device=dev)
    call('es_ground_focal', P(logits), T, B, Q, P(q2g), P(pos_map), P(gt_off), P(tlen), T, 0.25, 2.0, P(avg), 1.0, P(dlog), P(lsum), st)
    dbox = torch.zeros_like(boxes)
    lbox = torch.zeros(1, dtype=torch.float64, device=dev)
    call('es_box_cd_pairs', P(boxes), P(q2g), B, Q, P(gt_boxes), P(gt_off), n_pos, 1.0, farr([0.2, 0.2, 0.2, 0.4]), P(dbox), P(lbox), st)
    torch.cuda.synchronize()
    lc = float(lsum) / (float(avg) + float(torch.finfo(torch.float32).eps))
    print(f'loss_cls hip {lc:.7f} reference {float(d["loss_cls"]):.7f}; loss_bbox hip {float(lbox):.7f} reference {float(d["loss_bbox"]):.7f} (tol 1e-5)')
    assert abs(lc - float(d['loss_cls'])) < 1e-5 * max(1, abs(float(d['loss_cls'])))
    assert abs(float(lbox) - float(d['loss_bbox'])) < 1e-5 * max(1, abs(float(d['loss_bbox'])))
    e1, e2 = _rel(dlog, torch.from_numpy(d['dcls'][:, :, :T])), _rel(dbox, torch.from_numpy(d['dboxes']))
    print(f'gradients: d loss / d logits rel-L2 {e1:.2e}, d loss / d boxes {e2:.2e} (tol 1e-4)')
    assert e1 < 1e-4 and e2 < 1e-4
    # single-sample assign() protocol of the reference class
    from embodiedscan_amd.structures import InstanceData
    gi = asg.assign(InstanceData(scores_3d=logits.view(B, Q, T)[0], bboxes_3d=boxes.view(B, Q, 9)[0]),
                    InstanceData(bboxes_3d=gtb[0].to(dev), labels_3d=torch.zeros(Gs[0], dtype=torch.long),
                                 positive_maps=torch.from_numpy(d['pos_map0']).to(dev), text_token_mask=mask[0][None].repeat(Gs[0], 1)))
    np.testing.assert_array_equal(gi.cpu().numpy(), d['gt_inds0'])


# ----------------------------------------------------------------------------- model level
TEXT_CFG = dict(hidden_size=64, num_hidden_layers=1, num_attention_heads=2, intermediate_size=128, max_position_embeddings=64)


def _small_grounder(dev, num_layers=2, num_queries=32, thr=300, config='mv_grounding.py'):
    from embodiedscan_amd.config import build_detector, load_config
    cfg = load_config(os.path.join(ROOT, 'configs', config))
    m = cfg['model']
    m['num_queries'] = num_queries
    m['decoder']['num_layers'] = num_layers
    m['decoder']['layer_cfg']['ffn_cfg']['feedforward_channels'] = 128
    m['neck_3d']['pts_prune_threshold'] = thr
    m['text_encoder_cfg'] = TEXT_CFG
    det = build_detector(cfg, device=dev, seed=0).to(dev)
    # non-degenerate regression branch (the reference zero-initialises its last layer) and BN statistics
    g = torch.Generator().manual_seed(4)
    sd = {k: v.cpu() for k, v in det.state_dict().items()}
    for k in sd:
        if 'reg_branches' in k and k.endswith('.4.weight'):
            sd[k] = torch.randn(sd[k].shape, generator=g) * 0.05
        if 'reg_branches' in k and k.endswith('.4.bias'):
            sd[k] = torch.randn(sd[k].shape, generator=g) * 0.1
        if k.startswith('backbone.') and k.endswith('running_var'):
            sd[k] = torch.rand(sd[k].shape, generator=g) + 0.5
    for k in list(sd):                                   # shared branches: every alias must carry the same tensor
        if 'reg_branches.' in k and not k.startswith('bbox_head.reg_branches.0.'):
            sd[k] = sd['bbox_head.reg_branches.0.' + k.split('.', 3)[3]]
    det.load_state_dict({k: v.to(dev) for k, v in sd.items()})
    return cfg, det, sd


def _grounding_batch(dev, n=2, n_points=12000):
    from embodiedscan_amd import pipeline
    from embodiedscan_amd.synth import make_grounding_sample, make_scan
    scans = [make_scan(31 + i, n_views=3, height=120, width=160, img_size=(128, 128), n_points=n_points, n_boxes=8) for i in range(n)]
    anns = [make_grounding_sample(s, seed=i) for i, s in enumerate(scans)]
    dscans = [pipeline.upload_scan(s, dev) for s in scans]
    return scans, anns, dscans


def _hip_grounder_step(det, dscans, anns, mode, force=None):
    """one forward + backward of the small grounder on the HIP path; force = (query indices (B,Q), [per layer (B,Q) q2g]) from the oracle"""
    from embodiedscan_amd import engine as E, pipeline
    E.PRECISION[0] = mode
    det.force_queries = det.bbox_head.force_assign = None
    if force is not None:
        det.force_queries, det.bbox_head.force_assign = force
    try:
        E.WEIGHT_VERSION[0] += 1
        E.TAPE.clear()
        batch = pipeline.make_grounding_batch(dscans, anns)
        points_host = [p.cpu() for p in batch['inputs']['points']]
        data = det.data_preprocessor(batch, True)
        det._bind()
        det.arena.grad.zero_()
        E.new_grad_epoch()
        losses = det.forward(data['inputs'], data['data_samples'], mode='loss')
        out = dict(losses={k: float(v) for k, v in losses.items()}, points=points_host, data=data,
                   hid=[l['logits'].d.cpu() for l in det.bbox_head.last],
                   q2g=[l.get('q2g_free', l['q2g']).cpu() for l in det.bbox_head.last],
                   idx=(det.free_queries if force is not None else det.last_queries['idx']).cpu())
        det._backward(None)
        torch.cuda.synchronize()
        out['grads'] = {k: v.cpu() for k, v in det.arena.grad_dict().items()}
    finally:
        E.PRECISION[0] = 'f32'
        det.force_queries = det.bbox_head.force_assign = None
    return out


@pytest.mark.parametrize('mode', ['f32', 'bf16'])
def test_grounder_train_step_vs_oracle(dev, mode):
    """SparseFeatureFusion3DGrounder forward + backward (2 scans x 3 views, MinkNeck pruning live at 300 voxels, 32
    queries, 2 decoder layers) against the oracle.
    f32: query selection and Hungarian assignments identical; losses 1e-3, hidden states 1e-3, parameter gradients median 2e-3.
    bf16 (round 4: NEVER vacuous): the discrete decisions of a bf16 run -- the top-32 query boundary, a Hungarian near tie --
    may legitimately differ from the oracle's, after which nothing downstream is comparable.  So the free run only REPORTS them
    (with the cost margin of a flipped assignment on the oracle's own cost matrix, which must be a near tie), and a second,
    TEACHER-FORCED run takes the oracle's query indices and assignments (detector.force_queries / head.force_assign) and is
    held to the oracle's bf16-operand specification in every run: logits and losses 2e-2, parameter gradients median 2e-2 /
    90th percentile 1e-1 / worst 5e-1."""
    from oracle import grounding as OG, model as OM
    # MinkNeck pruning is live (300 voxels per level): the teacher-forced bf16 leg needs the pruned token lists of both sides to be
    # the same lists (asserted below: lengths and coordinates) so that the oracle's token indices address the same tokens
    thr = 300
    cfg, det, sd = _small_grounder(dev, thr=thr)
    scans, anns, dscans = _grounding_batch(dev)
    names = set(det.arena.grad_dict().keys())
    h = _hip_grounder_step(det, dscans, anns, mode)
    losses, hid, q2g, idx, grads, data, points_host = h['losses'], h['hid'], h['q2g'], h['idx'], h['grads'], h['data'], h['points']
    # ---- oracle on the same inputs; the frozen text encoder's output is an input of both paths
    th = det.last_text['hidden'].float().cpu()
    tmask = det.last_text['mask'].cpu()
    osd = {k: v.clone().requires_grad_(k in names) for k, v in sd.items()}
    imgs = torch.stack([OM.preprocess_img(torch.from_numpy(s['img']), MEAN, STD) for s in scans])
    gtb = [torch.from_numpy(a['gt_boxes']) for a in anns]
    pms = [ds.gt_instances_3d.positive_maps.cpu() for ds in data['data_samples']]
    # bf16 mode is compared with the oracle's bf16-OPERAND specification (oracle/rounding.py: every conv / Linear product
    # on rounded operands, f32 accumulation; the attention core and the contrastive logits stay f32 there, bf16 MFMA here)
    from contextlib import nullcontext
    from oracle import rounding as R
    with (R.bf16_operands() if mode == 'bf16' else nullcontext()):
        ol, aux = OG.grounder_loss(osd, points_host, imgs, [s['meta'] for s in scans], th, tmask, gtb, pms, num_queries=32,
                                   num_layers=2, thr=thr, return_aux=True)
        if mode == 'bf16':
            sum(ol.values()).backward()
    lens_h, lens_o = list(det.neck_3d.last['lens']), [int(f.shape[0]) for f in aux['feats_list']]
    assert lens_h == lens_o, (lens_h, lens_o)
    for b in range(2):                                   # token order: the oracle's index i is the HIP path's token i
        ch = det.neck_3d.last['points'].view(2, det.neck_3d.last['Lmax'], 3)[b, :lens_h[b]].cpu()
        assert float((ch - aux['coords'][b][:lens_h[b]]).abs().max()) < 1e-6
    same_q = torch.equal(idx.long(), aux['idx'])
    same_a = all(torch.equal((q2g[l][b] + 1).long(), aux['head'][l]['assign'][b]) for l in range(2) for b in range(2))
    print(f'{mode} free run: selected queries identical: {same_q}; Hungarian assignments identical: {same_a}')
    if mode == 'f32':
        assert same_q and same_a
    tol = 1e-3 if mode == 'f32' else 2e-2
    if mode == 'bf16':
        n_common = [len(set(idx[b].tolist()) & set(aux['idx'][b].tolist())) for b in range(2)]
        print(f'bf16 free run: {n_common} of 32 selected queries in common with the oracle')
        assert min(n_common) >= 24, 'more than a quarter of the queries differ: not a boundary effect'
        if same_q and not same_a:
            # WHICH margin flipped -- cost of our assignment minus the optimal one on the oracle's own cost matrix
            for l in range(2):
                for b in range(2):
                    if torch.equal((q2g[l][b] + 1).long(), aux['head'][l]['assign'][b]):
                        continue
                    G = gtb[b].shape[0]
                    gi, cost = OG.hungarian_assign(aux['head'][l]['cls'][b].detach(), aux['boxes'][l][b].detach(), gtb[b], pms[b],
                                                   tmask[b][None].repeat(max(G, 1), 1), return_cost=True)
                    mine = sum(float(cost[q, int(q2g[l][b][q])]) for q in torch.nonzero(q2g[l][b] >= 0).reshape(-1))
                    opt = sum(float(cost[q, int(gi[q]) - 1]) for q in torch.nonzero(gi > 0).reshape(-1))
                    print(f'bf16 layer {l} scan {b}: assignment differs, cost margin {mine - opt:.3e} on an optimal cost of {opt:.3e} (tol 2 %)')
                    assert mine - opt <= 2e-2 * max(abs(opt), 1.0)
        # the teacher-forced run: same discrete decisions as the oracle -> everything is comparable, in every run
        force = (aux['idx'].int(), [torch.stack([aux['head'][l]['assign'][b] - 1 for b in range(2)]).int() for l in range(2)])
        h = _hip_grounder_step(det, dscans, anns, mode, force=force)
        losses, hid, grads = h['losses'], h['hid'], h['grads']
        assert all(np.isfinite(v) for v in losses.values())
    T = tmask.shape[1]
    for l in range(2):
        ref = aux['head'][l]['cls'][:, :, :T].reshape(-1, T)
        keep = ~torch.isinf(ref)
        e = _rel(hid[l][keep], ref[keep].detach())
        print(f'{mode} decoder layer {l} token logits rel-L2 {e:.2e} (tol {tol:.0e})')
        assert e < tol
    for k in ol:
        e = abs(float(losses[k]) - float(ol[k])) / max(abs(float(ol[k])), 1e-6)
        print(f'{mode} {k}: hip {float(losses[k]):.6f} oracle {float(ol[k]):.6f} rel err {e:.2e} (tol {tol:.0e})')
        assert e < tol
    assert all(np.isfinite(float(v)) for v in losses.values()) and torch.isfinite(det.arena.grad).all()
    if mode == 'bf16':
        # the last bias of cross_posembed shifts every key of a sample by the same vector, which softmax cancels exactly: its true
        # gradient is zero and both sides hold rounding noise only (bf16 noise here: the 1e-6 norm threshold of the f32 leg does
        # not catch it) -- excluded by name, with its noise bounded against the weight of the same layer
        zero = 'decoder.cross_posembed.position_embedding_head.3.bias'
        assert float(grads[zero].norm()) < 5e-2 * float(grads[zero.replace('.bias', '.weight')].norm())
        rel = {k: _rel(v, osd[k].grad) for k, v in grads.items()
               if k != zero and osd[k].grad is not None and float(osd[k].grad.norm()) > 1e-6}
        near = ('decoder.', 'bbox_head.', 'text_feat_map.')          # behind the loss: no chaotic amplification yet
        grp = {'decoder / head / text map': [v for k, v in rel.items() if k.startswith(near)],
               'MinkNeck': [v for k, v in rel.items() if k.startswith('neck_3d.')],
               'backbones': [v for k, v in rel.items() if k.startswith(('backbone.', 'backbone_3d.'))]}
        v = np.sort(np.array(list(rel.values())))
        worst = max(rel, key=rel.get)
        print(f'bf16 (teacher-forced) gradients vs the bf16-operand oracle: {len(v)} tensors, median rel-L2 {float(np.median(v)):.2e}, 90th '
              f'percentile {float(v[int(0.9 * (len(v) - 1))]):.2e}, worst {rel[worst]:.2e} at {worst}; by distance from the loss: ' +
              '; '.join(f'{g} median {float(np.median(x)):.2e} worst {max(x):.2e} ({len(x)})' for g, x in grp.items()))
        # Gates: the tensors right behind the loss must agree tightly (the attention core -- bf16 P and V on the matrix cores here,
        # f32 in the oracle -- sets their floor); upstream, two bf16 summation orders drift apart through the train-mode BatchNorms
        # (tests/test_gpu_insitu.py holds every launch to 2e-4 on its own operands instead), so those only catch wiring errors
        assert len(v) > 100
        d = np.array(grp['decoder / head / text map'])
        # (measured on MI355X: decoder / head / text map median 3.6e-2 worst 1.1e-1; MinkNeck 4.6e-2 / 7.4e-2; backbones 1.1e-1 / 2.0e-1)
        assert float(np.median(d)) < 8e-2 and float(d.max()) < 3e-1, (float(np.median(d)), float(d.max()))
        assert float(np.median(v)) < 1.5e-1 and rel[worst] < 6e-1
    if mode == 'f32':
        sum(ol.values()).backward()
        # tensors whose true gradient is zero are skipped (norm < 1e-6): the last bias of cross_posembed shifts every key of a
        # sample by the same vector, which softmax cancels exactly -- both sides hold rounding noise only
        rel = {k: _rel(v, osd[k].grad) for k, v in grads.items() if osd[k].grad is not None and float(osd[k].grad.norm()) > 1e-6}
        worst = max(rel, key=rel.get)
        med = float(np.median(list(rel.values())))
        dec = [v for k, v in rel.items() if k.startswith(('decoder.', 'bbox_head.', 'text_feat_map.'))]
        print(f'f32 gradients vs oracle autograd: {len(rel)} tensors, median rel-L2 {med:.2e} (tol 2e-3), decoder/head/text median '
              f'{float(np.median(dec)):.2e}, worst {rel[worst]:.2e} at {worst} (tol 1e-1)')
        assert med < 2e-3 and float(np.median(dec)) < 1e-3 and rel[worst] < 1e-1


def test_grounder_train_and_predict(dev):
    """three optimiser steps (paramwise lr: decoder x0.1) lower nothing to NaN, and mode='predict' returns Q boxes with
    scores = sigmoid of the best token logit of the last layer, as the reference head does"""
    from embodiedscan_amd import engine as E, pipeline
    from embodiedscan_amd.config import build_optim_wrapper
    cfg, det, sd = _small_grounder(dev)
    scans, anns, dscans = _grounding_batch(dev)
    optim = build_optim_wrapper(cfg)
    E.PRECISION[0] = 'bf16'
    try:
        hist = []
        for _ in range(3):
            losses = det.train_step(pipeline.make_grounding_batch(dscans, anns), optim)
            hist.append(sum(float(v) for v in losses.values()))
        assert all(np.isfinite(hist)), hist
        assert [g[2] for g in optim.groups] == [1.0, 0.1, 1.0]
        batch = pipeline.make_grounding_batch(dscans, anns)
        data = det.data_preprocessor(batch, False)
        out = det.forward(data['inputs'], data['data_samples'], mode='predict')
    finally:
        E.PRECISION[0] = 'f32'
    r = out[0].pred_instances_3d
    assert r.bboxes_3d.tensor.shape == (32, 9) and r.scores_3d.shape == (32,)
    assert float(r.scores_3d.min()) >= 0 and float(r.scores_3d.max()) <= 1 and torch.isfinite(r.bboxes_3d.tensor).all()
    print(f'train 3 steps: total loss {hist[0]:.4f} -> {hist[-1]:.4f}; predict: {r.scores_3d.shape[0]} boxes, best score {float(r.scores_3d.max()):.4f}')


def test_grounder_with_fcaf_coder_vs_oracle(dev):
    """configs/mv_grounding_fcaf.py (= the reference's mv-grounding_..._fcaf-coder.py: box_coder='FCAF') shrunk like the test above,
    f32 mode: query selection and assignments identical to the oracle (whose FCAF coder is pinned to the reference's own,
    tests/test_oracle_golden.py), losses 1e-3, token logits 1e-3, parameter gradients median 2e-3 / worst 1e-1"""
    from oracle import grounding as OG, model as OM
    cfg, det, sd = _small_grounder(dev, config='mv_grounding_fcaf.py')
    assert det.bbox_head.box_coder == 'FCAF'
    scans, anns, dscans = _grounding_batch(dev)
    names = set(det.arena.grad_dict().keys())
    h = _hip_grounder_step(det, dscans, anns, 'f32')
    th, tmask = det.last_text['hidden'].float().cpu(), det.last_text['mask'].cpu()
    osd = {k: v.clone().requires_grad_(k in names) for k, v in sd.items()}
    imgs = torch.stack([OM.preprocess_img(torch.from_numpy(s['img']), MEAN, STD) for s in scans])
    gtb = [torch.from_numpy(a['gt_boxes']) for a in anns]
    pms = [ds.gt_instances_3d.positive_maps.cpu() for ds in h['data']['data_samples']]
    ol, aux = OG.grounder_loss(osd, h['points'], imgs, [s['meta'] for s in scans], th, tmask, gtb, pms, num_queries=32, num_layers=2,
                               thr=300, return_aux=True, coder='FCAF')
    sum(ol.values()).backward()
    assert torch.equal(h['idx'].long(), aux['idx'])
    assert all(torch.equal((h['q2g'][l][b] + 1).long(), aux['head'][l]['assign'][b]) for l in range(2) for b in range(2))
    T = tmask.shape[1]
    for l in range(2):
        ref = aux['head'][l]['cls'][:, :, :T].reshape(-1, T)
        keep = ~torch.isinf(ref)
        assert _rel(h['hid'][l][keep], ref[keep].detach()) < 1e-3
        e = _rel(det.bbox_head.last[l]['boxes'].d, aux['boxes'][l].detach().reshape(-1, 9))
        print(f'FCAF coder, decoder layer {l}: decoded boxes rel-L2 {e:.2e} (tol 1e-4)')
        assert e < 1e-4
    for k in ol:
        e = abs(h['losses'][k] - float(ol[k])) / max(abs(float(ol[k])), 1e-6)
        print(f'FCAF coder {k}: hip {h["losses"][k]:.6f} oracle {float(ol[k]):.6f} rel err {e:.2e} (tol 1e-3)')
        assert e < 1e-3
    rel = {k: _rel(v, osd[k].grad) for k, v in h['grads'].items() if osd[k].grad is not None and float(osd[k].grad.norm()) > 1e-6}
    worst = max(rel, key=rel.get)
    med = float(np.median(list(rel.values())))
    print(f'FCAF coder, f32 gradients vs oracle autograd: {len(rel)} tensors, median rel-L2 {med:.2e} (tol 2e-3), worst {rel[worst]:.2e} at {worst} (tol 1e-1)')
    assert med < 2e-3 and rel[worst] < 1e-1
