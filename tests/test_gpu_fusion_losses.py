"""A8 / A13-A15 kernels against the golden vectors made by the reference's own code and against the oracle."""
import os
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _meta_from(d):
    meta = dict(img_shape=tuple(int(v) for v in d['img_shape']), scale_factor=tuple(float(v) for v in d['scale_factor']),
                img_crop_offset=tuple(float(v) for v in d['crop_offset']), flip=bool(d['flip']))
    flow = [str(x) for x in d['flow']]
    if flow:
        meta.update(pcd_rotation=d['pcd_rotation'], pcd_scale_factor=float(d['pcd_scale_factor']), pcd_trans=d['pcd_trans'],
                    pcd_horizontal_flip=bool(d['hflip']), pcd_vertical_flip=bool(d['vflip']), transformation_3d_flow=flow)
    return meta


def test_point_sample_vs_reference_golden(golden_dir):
    """The golden points are arbitrary floats, the kernel takes integer voxel coordinates * voxel_size, so the check
    runs on the quantised points through BOTH the kernel and the oracle (itself pinned to the golden file)."""
    from embodiedscan_amd import engine as E
    from embodiedscan_amd.hip import P, call
    from embodiedscan_amd.models.layers.fusion_layers.point_fusion import build_fusion_meta
    from oracle import model as OM
    dev = torch.device('cuda:0')
    for name in ('point_sample_plain', 'point_sample_aug'):
        d = np.load(os.path.join(golden_dir, name + '.npz'))
        meta = _meta_from(d)
        V = d['proj'].shape[0]
        # express the projection as intrinsic @ extrinsic with identity intrinsics
        meta['depth2img'] = dict(intrinsic=[np.eye(4, dtype=np.float32)] * V, extrinsic=[d['proj'][v] for v in range(V)])
        vs = 0.01
        ci = np.round(d['points'] / vs).astype(np.int32)
        coords = np.concatenate([np.zeros((len(ci), 1), np.int32), ci], 1)
        pts_q = torch.from_numpy(ci).float() * vs
        feats = torch.from_numpy(d['feats'])                       # (V,C,H,W)
        ref = OM.batch_point_sample(meta, feats, pts_q, torch.from_numpy(d['proj']), torch.from_numpy(d['scale_factor']),
                                    torch.from_numpy(d['crop_offset']), bool(d['flip']), tuple(d['pad_shape']),
                                    tuple(d['img_shape']))
        Vn, C, H, W = feats.shape
        nhwc = feats.permute(0, 2, 3, 1).contiguous().reshape(Vn * H * W, C).to(dev)
        md = build_fusion_meta([meta], 'DEPTH', tuple(int(v) for v in d['pad_shape']), V).to(dev)
        out = torch.zeros((len(ci), C), device=dev)
        pix = torch.empty((len(ci), V), dtype=torch.int32, device=dev)
        cnt = torch.empty(len(ci), dtype=torch.int32, device=dev)
        cd = torch.from_numpy(coords).to(dev)
        call('es_point_sample_fwd', P(cd), len(ci), vs, P(md), md.shape[1], V, P(nhwc), H, W,
             C, P(out), C, P(pix), P(cnt), torch.cuda.current_stream().cuda_stream)
        diff = (out.cpu() - ref).abs().max(1).values
        bad = int((diff > 1e-5).sum())
        print(f'{name}: rows differing from the oracle: {bad}/{len(ci)} (allowed: 0.5% -- nearest-pixel rounding ties)')
        assert bad <= max(1, len(ci) // 200)
        assert (ref != 0).any(1).sum() > 20


def test_losses_vs_oracle(golden_dir):
    from embodiedscan_amd.hip import P, call, farr, iarr, parr
    from oracle import geometry as G
    dev = torch.device('cuda:0')
    d = np.load(os.path.join(golden_dir, 'box_coder_cdloss.npz'))
    n = d['pred'].shape[0]
    g = torch.Generator().manual_seed(3)
    pts, pred, tgt = torch.from_numpy(d['points']), torch.from_numpy(d['pred']), torch.from_numpy(d['target'])
    cls_t = torch.where(torch.rand(n, generator=g) < 0.7, torch.randint(0, 284, (n,), generator=g), torch.tensor(-1)).int()
    center_p, center_t = torch.randn(n, generator=g), torch.rand(n, generator=g)
    npos = int((cls_t >= 0).sum())
    avg = torch.tensor([float(max(npos, 1))])
    w = [0.2, 0.2, 0.2, 0.4]
    # oracle value + autograd gradient
    po = pred.clone().requires_grad_(True)
    co = center_p.clone().requires_grad_(True)
    pos = torch.nonzero(cls_t >= 0).squeeze(1)
    dec = G.bbox_pred_to_bbox(pts[pos], po[pos])
    tb = tgt[pos]
    lb = w[0] * G.bbox_cd_loss(torch.cat((dec[:, :3], tb[:, 3:]), -1), tb) + \
        w[1] * G.bbox_cd_loss(torch.cat((tb[:, :3], dec[:, 3:6], tb[:, 6:]), -1), tb) + \
        w[2] * G.bbox_cd_loss(torch.cat((tb[:, :6], dec[:, 6:]), -1), tb) + w[3] * G.bbox_cd_loss(dec, tb)
    lc = torch.nn.functional.binary_cross_entropy_with_logits(co[pos], center_t[pos], reduction='sum') / (avg[0] + 1.1920929e-07)
    (lb + lc).backward()
    dcen = torch.zeros(n, device=dev)
    dbb = torch.zeros((n, 12), device=dev)
    acc = torch.zeros(2, dtype=torch.float64, device=dev)          # f64 loss sums
    # keep every device tensor alive in a variable: P() only takes the pointer
    d_cls, d_np, d_pts, d_cp = cls_t.to(dev), torch.tensor([npos], dtype=torch.int32, device=dev), pts.to(dev), center_p.to(dev)
    d_pred, d_ct, d_tgt, d_avg = pred.to(dev), center_t.to(dev), tgt.to(dev), avg.to(dev)
    # two "levels" (rows 0..39 and 40..n) to exercise the level table
    n0 = 40
    d_ws = torch.empty(n + 1, dtype=torch.int32, device=dev)     # compacted positives (bound: every row)
    call('es_pos_losses', P(d_cls), n, P(d_np), n, P(d_ws), P(d_pts), 2, iarr([0, n0, n]), parr([d_cp.data_ptr(), d_cp.data_ptr() + 4 * n0]),
         parr([d_pred.data_ptr(), d_pred.data_ptr() + 48 * n0]), parr([dcen.data_ptr(), dcen.data_ptr() + 4 * n0]),
         parr([dbb.data_ptr(), dbb.data_ptr() + 48 * n0]), 1, P(d_ct), P(d_tgt), P(d_avg), 1.0, farr(w), P(acc),
         torch.cuda.current_stream().cuda_stream)
    acc = acc.cpu()
    lb, lc = lb.detach(), lc.detach()
    print(f'bbox loss hip {float(acc[1]):.6f} oracle {float(lb):.6f}; center sum hip {float(acc[0]):.6f}')
    assert abs(float(acc[1]) - float(lb)) / float(lb) < 1e-5
    assert abs(float(acc[0]) / (float(avg[0]) + 1.1920929e-07) - float(lc)) / float(lc) < 1e-5
    e1 = float((dbb.cpu() - po.grad).abs().max() / po.grad.abs().max())
    e2 = float((dcen.cpu() - co.grad).abs().max() / co.grad.abs().max())
    print(f'dual-number box-loss gradient rel err {e1:.2e}, centerness gradient rel err {e2:.2e} (tol 1e-4)')
    assert e1 < 1e-4 and e2 < 1e-4
    # focal
    N, C = 3000, 284
    logits = torch.randn(N, C, generator=g) * 2 - 2
    labels = torch.where(torch.rand(N, generator=g) < 0.1, torch.randint(0, C, (N,), generator=g), torch.tensor(-1))
    lo = logits.clone().requires_grad_(True)
    fl = G.sigmoid_focal_loss_sum(lo, labels) / (avg[0] + 1.1920929e-07)
    fl.backward()
    grad = torch.zeros((N, C), device=dev)
    partial = torch.empty(2048, dtype=torch.float64, device=dev)
    out = torch.zeros(1, device=dev)
    d_log, d_lab = logits.to(dev), labels.int().to(dev)
    call('es_focal_loss', P(d_log), C, P(d_lab), N, C, 2.0, 0.25, P(d_avg), 1.0, P(grad), C,
         P(partial), P(out), torch.cuda.current_stream().cuda_stream)
    e = abs(float(out.cpu()) - float(fl)) / float(fl)
    eg = float((grad.cpu() - lo.grad).abs().max() / lo.grad.abs().max())
    print(f'focal loss rel err {e:.2e} grad rel err {eg:.2e} (tol 1e-5 / 1e-4)')
    assert e < 1e-5 and eg < 1e-4


def test_optimizer_step():
    from embodiedscan_amd.hip import P, call
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(0)
    n = 100003
    p, gr = torch.randn(n, generator=g), torch.randn(n, generator=g) * 3
    pt = p.clone().requires_grad_(True)
    opt = torch.optim.AdamW([pt], lr=1e-3, weight_decay=1e-4)
    pd, gd = p.to(dev), gr.to(dev)
    m, v = torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    partial, norm = torch.empty(2048, dtype=torch.float64, device=dev), torch.zeros(1, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    for step in range(1, 4):
        pt.grad = gr.clone()
        torch.nn.utils.clip_grad_norm_([pt], 10.0)
        opt.step()
        call('es_grad_norm', P(gd), n, P(partial), P(norm), s)
        call('es_adamw_step', P(pd), P(gd), P(m), P(v), n, 1e-3, 0.9, 0.999, 1e-8, 1e-4, step, 10.0, P(norm), 1.0, s)
    assert abs(float(norm.cpu()) - float(gr.norm())) / float(gr.norm()) < 1e-6
    err = float((pd.cpu() - pt.detach()).abs().max())
    print(f'AdamW 3 steps max abs err {err:.2e} (tol 1e-6)')
    assert err < 1e-6


def test_point_sample_bwd_is_an_ordered_gather():
    """es_point_sample_bwd (round 3: linked hit lists + one wave per feature-map pixel adding its hits in ascending voxel
    order) against index_add_ in f64: pixels with > 64 and > 128 hits (the multi-round extraction), voxels WITHOUT a valid
    view whose pix entries are >= 0 all the same (the forward samples unmasked, SURVEY Q3: they must not contribute),
    empty pixels written as zeros, accumulate on / off; two runs bit-identical."""
    from embodiedscan_amd.hip import P, call
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(3)
    B, V, Hf, Wf, C, n = 2, 3, 6, 5, 96, 4000
    HW = Hf * Wf
    coords = torch.zeros((n, 4), dtype=torch.int32)
    coords[:, 0] = (torch.arange(n) >= n // 2).int()
    pix = torch.randint(-1, HW, (n, V), generator=g, dtype=torch.int32)
    pix[:600, 0] = 7                                         # 600 hits on one pixel of (sample 0, view 0)
    pix[torch.rand(n, V, generator=g) < 0.3] = -1
    cnt = (pix >= 0).sum(1).int()
    dead = torch.rand(n, generator=g) < 0.1                  # sampled somewhere, but no VALID view
    cnt[dead] = 0
    dout = torch.randn(n, C + 8, generator=g)                # strided rows (columns 8.. are the image part)
    want = torch.zeros(B * V * HW, C, dtype=torch.float64)
    for v in range(V):
        m = (pix[:, v] >= 0) & (cnt > 0)
        rows = (coords[m, 0].long() * V + v) * HW + pix[m, v].long()
        want.index_add_(0, rows, dout[m, 8:].double() / cnt[m, None].double())
    d = dict(coords=coords.to(dev), pix=pix.to(dev), cnt=cnt.to(dev), dout=dout.to(dev))
    head = torch.empty(B * V * HW, dtype=torch.int32, device=dev)
    nxt = torch.empty(n * V, dtype=torch.int32, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    outs = []
    for _ in range(2):
        f = torch.full((B * V * HW, C), float('nan'), device=dev)        # accumulate = 0 must overwrite everything
        call('es_point_sample_bwd', P(d['coords']), n, V, d['dout'].data_ptr() + 32, C + 8, P(d['pix']), P(d['cnt']), Hf, Wf, C,
             P(f), B * V, P(head), P(nxt), 0, st)
        outs.append(f.clone())
    assert torch.equal(outs[0], outs[1])
    e = float((outs[0].double().cpu() - want).norm() / want.norm())
    print(f'point_sample_bwd vs f64 index_add: rel-L2 {e:.2e} (tol 1e-6); busiest pixel {int((pix[:, 0] == 7).sum())} hits')
    assert e < 1e-6 and torch.isfinite(outs[0]).all()
    base = torch.randn(B * V * HW, C, generator=g)
    f = base.to(dev)
    call('es_point_sample_bwd', P(d['coords']), n, V, d['dout'].data_ptr() + 32, C + 8, P(d['pix']), P(d['cnt']), Hf, Wf, C,
         P(f), B * V, P(head), P(nxt), 1, st)
    assert float((f.double().cpu() - (want + base.double())).norm() / want.norm()) < 1e-6
