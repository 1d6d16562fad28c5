"""In-situ check of EVERY backward launch with matrix-core arithmetic or parameters, of a real bf16 train step of ALL THREE
detectors, against its arithmetic specification ("generalise over the grounder and the occupancy
detector ... every launch <= 2e-4 on the operands it saw").

End to end, a bf16 implementation cannot be held to its specification more tightly than to f32: rounding is discontinuous,
so two summation orders that agree to 1e-6 on one layer put ~2.5e-4 of the next layer's inputs on different sides of a bf16
rounding boundary, and after four or five layers the activations differ by the full bf16 quantisation noise (measured:
feature maps 6e-4 -> 2.4e-3 -> 5.3e-3 by depth, parameter gradients 10 % median from a COMMON head-output gradient; through the
occupancy neck's train-mode BatchNorms over 4 .. 256 rows 37 % median).  What CAN be held tightly is every launch on the
operands it actually saw.  engine.DEBUG_CONV records, for each convolution / Linear backward of the step -- sparse 3-D (27 / 1
taps, strided), the dense 3-D neck of the occupancy detector (768 .. 3072 channels at config-5 scale: the huge weight-gradient
tile and the LDS-DMA kernel on REAL operands), the image backbone's fused conv + BN + ReLU layers with their gated data
gradients and bf16 activation rows, the fused 8-tap generative transposed convolutions, head GEMMs, the decoder's K = 1
256 -> 256 Linear layers and attention in-projections -- the input rows, the output gradient, the map, and the data gradient the
launch produced; weight and bias gradients are read from the arena after the step.  engine.DEBUG_OPS records the attention,
LayerNorm and ContrastiveEmbed backward launches of the grounder.

Specifications (oracle/rounding.py restated on raw tensors; r = round-to-nearest-even bf16 for layers with >= 16 input channels
(data gradient: >= 16 output channels), f32 accumulate):
  convolution:  dw[k] = r(x[nbr[:, k]])^T r(gy),  dx = sum_k scatter(r(gy) r(w[k])^T)  (* BN scale and ReLU mask for a gated
                launch),  db = column sums of gy
  attention  :  P = exp(r(q s) r(k)^T - lse), dV = r(P)^T r(dO), dS = P (r(dO) r(V)^T - delta), dQ = s r(dS) r(K),
                dK = r(dS)^T r(q s)   (s = 1 / sqrt(32); lse / delta as the forward / the delta kernel wrote them)
  LayerNorm, ContrastiveEmbed: exact f32 formulas in f64.
Tolerance 2e-4 relative L2 per launch (only the f32 summation order -- and, for attention, the fast exponential flipping a
few bf16 roundings -- is left; measured ~1e-6 .. 6e-5).  Small cases are specified on the host (CPU, f64); the config-4 /
config-5 scale cases evaluate the same formulas with torch in f64 on the GPU (rocBLAS -- an implementation independent of the
kernels under test), because the 3072^2 x 27 layers would take minutes on host cores."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOL = 2e-4


def _rel(a, b):
    a, b = a.double(), b.double().to(a.device)
    return float((a - b).norm() / (b.norm() + 1e-30))


def _r(t):
    return t.to(torch.bfloat16).to(torch.float64)


def _spec(x, w, nbr, n_out, gy, round_fwd, round_dgrad, gate, xact):
    """-> (dw (K,cin,cout) f64, dx (n_in,cin) f64) of one convolution backward (all operands on one device)"""
    K, cin, cout = w.shape
    n_in = x.shape[0]
    xw = _r(x) if round_fwd else x.double()
    gw = _r(gy) if round_fwd else gy.double()                 # weight gradient: both operands rounded in bf16 mode
    gd = _r(gy) if round_dgrad else gy.double()
    wd = _r(w) if round_dgrad else w.double()
    dw = torch.zeros((K, cin, cout), dtype=torch.float64, device=x.device)
    dx = torch.zeros((n_in, cin), dtype=torch.float64, device=x.device)
    for k in range(K):
        if nbr is None:
            rows = torch.arange(min(n_out, n_in), device=x.device)
            src = rows
        else:
            rows = torch.nonzero(nbr[:, k] >= 0).squeeze(1)
            src = nbr[rows, k].long()
        if rows.numel() == 0:
            continue
        dw[k] = xw[src].t() @ gw[rows]
        dx.index_add_(0, src, gd[rows] @ wd[k].t())
    if gate is not None:                                        # fused ReLU mask + frozen-BN scale of the producer of x
        dx = dx * gate.double()[None, :] * (xact.double() > 0)
    return dw, dx


def _check_conv_records(recs, label, dev):
    """every weight / bias / data gradient of the recorded convolution backwards against _spec; dev: where the
    specification is evaluated (cpu or the GPU, f64 either way)"""
    to = lambda t: None if t is None else t.to(dev)
    by_w, by_b = {}, {}
    for r in recs:
        if r['w'].g is not None:
            by_w.setdefault(r['w'].g.data_ptr(), []).append(r)
        if r.get('bias') is not None and r['bias'].g is not None:
            by_b.setdefault(r['bias'].g.data_ptr(), []).append(r)
    n_dw = n_dx = n_db = 0
    worst_dw = worst_dx = worst_db = (0.0, '')
    kinds = set()
    groups = {}
    seen = set()
    for ptr, rs in by_w.items():
        w = rs[0]['w']
        K, cin, cout = w.d.shape
        wh = to(w.d.float())
        dw_sum = torch.zeros((K, cin, cout), dtype=torch.float64, device=dev)
        for r in rs:
            seen.add(id(r))
            x, gy = to(r['x'].float()), to(r['gy'].float())
            nbr = to(r['nbr'])
            bf = r['bf']
            gate = None if r['gate'] is None else to(r['gate'].float())
            dw, dx = _spec(x, wh, nbr, r['n_out'], gy, bf, bf and (cout >= 16 or gate is not None), gate, x)
            dw_sum += dw
            tag = f'K={K} {cin}->{cout} rows {x.shape[0]}->{r["n_out"]}' + (' gated' if gate is not None else '') + \
                  (' bf16-rows' if r['x'].dtype == torch.bfloat16 else '') + ('' if bf else ' exact-f32') + \
                  (' gen-tap' if 'group' in r else '')
            kinds.add((K, cin, cout, gate is not None, r['x'].dtype == torch.bfloat16, bf, 'group' in r))
            if 'group' in r:                                     # 8 taps of a generative transposed conv: ONE data gradient
                g = groups.setdefault(r['group'], dict(dx=torch.zeros_like(dx), n=0, tag=tag, before=r['before'], after=None,
                                                      need=r['need_dx']))
                g['dx'] += dx
                g['n'] += 1
                if 'after' in r:
                    g['after'] = r['after']
                continue
            if r['need_dx'] and float(dx.norm()) > 0:
                # the launch either wrote the buffer or ACCUMULATED onto what another consumer of x had put there: compare
                # after with before + specification; the f32 rounding of that sum is legitimately eps * |after|
                after = to(r['after']).double()
                want = dx if r['before'] is None else to(r['before']).double() + dx
                e = float((after - want).norm() / dx.norm())
                tol = TOL + 4e-7 * float(after.norm() / dx.norm())
                n_dx += 1
                if e > worst_dx[0]:
                    worst_dx = (e, f'{tag} (tol applied {tol:.1e})')
                assert e < tol, f'{label}: data gradient of {tag}: rel-L2 {e:.2e} (tol {tol:.1e})'
        if float(dw_sum.norm()) > 0:
            e = _rel(to(w.g), dw_sum)
            n_dw += 1
            if e > worst_dw[0]:
                worst_dw = (e, f'K={K} {cin}->{cout} ({len(rs)} launch(es))')
            assert e < TOL, f'{label}: weight gradient K={K} {cin}->{cout} ({len(rs)} launches): rel-L2 {e:.2e} (tol {TOL:.0e})'
    for g in groups.values():
        assert g['n'] == 8
        if g['need'] and float(g['dx'].norm()) > 0:
            after = to(g['after']).double()
            want = g['dx'] if g['before'] is None else to(g['before']).double() + g['dx']
            e = float((after - want).norm() / g['dx'].norm())
            tol = TOL + 4e-7 * float(after.norm() / g['dx'].norm())
            n_dx += 1
            if e > worst_dx[0]:
                worst_dx = (e, f'{g["tag"]} x 8 (tol applied {tol:.1e})')
            assert e < tol, f'{label}: data gradient of the generative transposed conv {g["tag"]}: rel-L2 {e:.2e} (tol {tol:.1e})'
    for ptr, rs in by_b.items():
        b = rs[0]['bias']
        bf_ = rs[0]['bias_from']
        want = sum(to(r['gy'].float()).double()[:r['n_out'], bf_:].sum(0) for r in rs)
        mag = sum(to(r['gy'].float()).double()[:r['n_out'], bf_:].abs().sum(0) for r in rs)
        if float(want.norm()) > 0:
            # a column sum may cancel almost completely (the bias of an attention KEY projection: softmax is invariant to a shift
            # of all keys, sum_j dK_j = 0 in exact arithmetic): an f32 sum is then only good to eps * sum |gy| -- the tolerance says so
            e = _rel(to(b.g)[bf_:], want)
            tol = TOL + 1e-5 * float(mag.norm() / want.norm())
            n_db += 1
            if e > worst_db[0]:
                worst_db = (e, f'{tuple(b.g.shape)} from column {bf_} ({len(rs)} launch(es), tol applied {tol:.1e})')
            assert e < tol, f'{label}: bias gradient {tuple(b.g.shape)}: rel-L2 {e:.2e} (tol {tol:.1e})'
    print(f'{label}: {len(recs)} convolution / Linear backwards, {len(kinds)} launch classes: {n_dw} weight gradients, worst rel-L2 '
          f'{worst_dw[0]:.2e} at {worst_dw[1]}; {n_dx} data gradients, worst {worst_dx[0]:.2e} at {worst_dx[1]}; {n_db} bias '
          f'gradients, worst {worst_db[0]:.2e} at {worst_db[1]} (tol {TOL:.0e} + eps-of-accumulated-buffer)')
    return n_dw, n_dx, n_db, kinds


def _check_ops(ops, label, dev):
    """attention / LayerNorm / ContrastiveEmbed backward launches against their specifications"""
    to = lambda t: None if t is None else t.to(dev)
    n = dict(attn=0, ln=0, contrastive=0)
    worst = dict(attn=(0.0, ''), ln=(0.0, ''), contrastive=(0.0, ''))

    def note(kind, e, tag, tol=TOL):
        if e > worst[kind][0]:
            worst[kind] = (e, tag)
        assert e < tol, f'{label}: {kind} backward, {tag}: rel-L2 {e:.2e} (tol {tol:.0e})'
    ln_w = {}
    for r in ops:
        k = r['kind']
        n[k] += 1
        if k == 'attn':
            B, H, Lq, Lk, bf = r['B'], r['H'], r['Lq'], r['Lk'], r['bf']
            s = 1.0 / 32 ** 0.5
            rr = _r if bf else (lambda t: t.double())
            hd = lambda t, L: to(t.float()).view(B, L, H, 32).permute(0, 2, 1, 3)
            q, kk, v, o, do = hd(r['q'], Lq), hd(r['k'], Lk), hd(r['v'], Lk), hd(r['o'], Lq), hd(r['do'], Lq)
            qs, ks, vs, dos = rr(q * s), rr(kk), rr(v), rr(do)             # (q * s) is formed in f32, then rounded
            lse = to(r['lse']).double().view(B, H, Lq)
            delta = to(r['delta']).double().view(B, H, Lq)
            want_delta = (do.double() * o.double()).sum(-1)
            note('attn', _rel(delta, want_delta), f'delta B={B} Lq={Lq}', 1e-5)
            klen = torch.full((B,), Lk, device=dev) if r['klen'] is None else to(r['klen']).long().clamp(max=Lk)
            live = (torch.arange(Lk, device=dev)[None, :] < klen[:, None])[:, None, None, :]
            Pm = torch.exp(qs @ ks.transpose(-1, -2) - lse[..., None]) * live
            dS = Pm * (dos @ vs.transpose(-1, -2) - delta[..., None])
            Pr, dSr = rr(Pm.float()), rr(dS.float())
            back = lambda t, L: t.permute(0, 2, 1, 3).reshape(B * L, H * 32)
            want = [back(s * (dSr @ ks), Lq), back(dSr.transpose(-1, -2) @ qs, Lk), back(Pr.transpose(-1, -2) @ dos, Lk)]
            for name, got, w_, i in (('dq', r['dq'], want[0], 0), ('dk', r['dk'], want[1], 1), ('dv', r['dv'], want[2], 2)):
                if r['acc']:
                    w_ = w_ + to(r['before'][i]).double()
                note('attn', _rel(to(got), w_), f'{name} B={B} H={H} Lq={Lq} Lk={Lk} bf16={bf}')
        elif k == 'ln':
            dy, z, w = to(r['dy']).double(), to(r['z']).double(), to(r['w']).double()
            mean, rstd = to(r['mean']).double()[:, None], to(r['rstd']).double()[:, None]
            xh = (z - mean) * rstd
            g = dy * w
            dz = rstd * (g - g.mean(1, keepdim=True) - xh * (g * xh).mean(1, keepdim=True))
            note('ln', _rel(to(r['dz']), dz), f'dz n={r["n"]} C={r["C"]}', 1e-5)
            # parameter gradients accumulate over the launches that share the LayerNorm: check each launch's increment
            tol = lambda inc, tot: 1e-5 + 4e-7 * float(tot.norm() / (inc.norm() + 1e-30))
            for nm, a0, a1, want in (('dw', r['dw0'], r['dw1'], (dy * xh).sum(0)), ('db', r['db0'], r['db1'], dy.sum(0))):
                inc = to(a1).double() - to(a0).double()
                if float(want.norm()) > 0:
                    note('ln', _rel(inc, want), f'{nm} n={r["n"]}', max(tol(want, to(a1).double()), 2e-5))
        elif k == 'contrastive':
            B, L, T, C = r['B'], r['L'], r['T'], r['C']
            dl, v, text = to(r['dl']).double().view(B, L, T), to(r['v']).double().view(B, L, C), to(r['text']).double().view(B, T, C)
            tl = to(r['tlen']).long()
            live = (torch.arange(T, device=dev)[None, :] < tl[:, None])[:, None, :]
            dl = dl * live
            inv = 1.0 / C ** 0.5
            if r['dv1'] is not None:
                want = (dl @ text) * inv
                if r['acc']:
                    want = want + to(r['dv0']).double().view(B, L, C)
                note('contrastive', _rel(to(r['dv1']).view(B, L, C), want), f'dv B={B} L={L} T={T}', 1e-5)
            if r['dtext1'] is not None:
                inc = to(r['dtext1']).double().view(B, T, C) - to(r['dtext0']).double().view(B, T, C)
                want = (dl.transpose(1, 2) @ v) * inv
                tol = 1e-5 + 4e-7 * float(to(r['dtext1']).double().norm() / (want.norm() + 1e-30))
                note('contrastive', _rel(inc, want), f'dtext B={B} L={L} T={T}', max(tol, 2e-5))
            inc = float(to(r['dbias1']).double().sum() - to(r['dbias0']).double().sum())
            wb = float(dl.sum())
            if abs(wb) > 0:
                e = abs(inc - wb) / abs(wb)
                note('contrastive', e, f'dbias B={B}', 1e-4 + 1e-6 * abs(float(to(r['dbias1']).double().sum())) / abs(wb))
    print(f'{label}: ' + '; '.join(f'{n[k]} {k} backward launches, worst rel-L2 {worst[k][0]:.2e} at {worst[k][1]}' for k in n if n[k]))
    return n


def _recorded_step(det, make_batch, backward):
    """one bf16 forward + backward of `det` with every backward launch recorded"""
    from embodiedscan_amd import engine as E
    E.PRECISION[0] = 'bf16'
    E.DEBUG_CONV, E.DEBUG_OPS = [], []
    try:
        E.TAPE.clear()
        E.WEIGHT_VERSION[0] += 1
        batch = make_batch()
        data = det.data_preprocessor(batch, True)
        det._bind()
        det.arena.grad.zero_()
        E.new_grad_epoch()
        if hasattr(det, '_tape_parts'):
            det._tape_parts = []
        losses = det.forward(data['inputs'], data['data_samples'], mode='loss')
        backward()
        torch.cuda.synchronize()
        recs, ops = E.DEBUG_CONV, E.DEBUG_OPS
    finally:
        E.DEBUG_CONV = E.DEBUG_OPS = None
        E.PRECISION[0] = 'f32'
    assert all(np.isfinite(float(v)) for v in losses.values())
    return recs, ops


def test_every_conv_backward_of_a_bf16_step_matches_its_specification():
    """mv-3ddet (configs/mv_3ddet.py), 2 scans x 3 views"""
    from embodiedscan_amd import pipeline
    from embodiedscan_amd.config import build_detector
    from embodiedscan_amd.synth import make_scan
    dev = torch.device('cuda:0')
    det = build_detector(os.path.join(ROOT, 'configs', 'mv_3ddet.py'), device=dev, seed=0).to(dev)
    scans = [make_scan(s, n_views=3, height=240, width=320, img_size=(192, 192), n_points=15000) for s in (21, 22)]
    dscans = [pipeline.upload_scan(s, dev) for s in scans]
    recs, ops = _recorded_step(det, lambda: pipeline.make_batch(dscans), lambda: det._backward(None))
    assert len(recs) > 80, len(recs)
    n_dw, n_dx, n_db, kinds = _check_conv_records(recs, 'mv-3ddet', torch.device('cpu'))
    assert n_dw > 60 and n_dx > 60 and n_db >= 1
    assert any(k[6] for k in kinds), 'the fused generative transposed convolutions were not recorded'


def test_every_backward_launch_of_a_bf16_grounder_step_matches_its_specification():
    """SparseFeatureFusion3DGrounder (configs/mv_grounding.py shrunk: 2 scans x 3 views, 32 queries, 2 decoder layers, MinkNeck
    pruning live): convolutions + the decoder's Linear layers / in-projections, attention, LayerNorm, ContrastiveEmbed"""
    import test_gpu_grounding as TG
    from embodiedscan_amd import pipeline
    dev = torch.device('cuda:0')
    cfg, det, sd = TG._small_grounder(dev)
    scans, anns, dscans = TG._grounding_batch(dev)
    recs, ops = _recorded_step(det, lambda: pipeline.make_grounding_batch(dscans, anns), lambda: det._backward(None))
    n_dw, n_dx, n_db, kinds = _check_conv_records(recs, 'mv-grounding (small)', torch.device('cpu'))
    assert n_dw > 90 and n_dx > 90 and n_db > 10
    assert any(k[0] == 1 and k[1] == 256 and k[2] == 256 for k in kinds), 'no K = 1 256 -> 256 decoder launch was recorded'
    n = _check_ops(ops, 'mv-grounding (small)', torch.device('cpu'))
    assert n['attn'] == 2 * 3 and n['ln'] >= 2 * 4 and n['contrastive'] == 2       # per decoder layer: self, text, point attention


def test_every_conv_backward_of_a_bf16_occupancy_step_matches_its_specification():
    """DenseFusionOccPredictor (configs/mv_occ.py shrunk: 8 x 8 x 4 volume, 3 views, width-16 backbones)"""
    import test_gpu_occ as TO
    from embodiedscan_amd import engine as E, pipeline
    dev = torch.device('cuda:0')
    cfg = TO._small_cfg()
    det, scan, occ, dscan = TO._occ_case(dev, cfg)
    recs, ops = _recorded_step(det, lambda: pipeline.make_occ_batch([dscan], [occ]), lambda: E.TAPE.backward())
    n_dw, n_dx, n_db, kinds = _check_conv_records(recs, 'occupancy (small)', torch.device('cpu'))
    assert n_dw > 80 and n_dx > 80


def test_config5_scale_occupancy_step_in_situ():
    """the SHIPPED occupancy detector (ResNet-50 + FPN 256, neck 768 -> 1536 -> 3072, 40 x 40 x 16 voxels, 10 views, 751 M
    parameters): every convolution backward of one bf16 step on its real operands -- k_spconv_wgrad_bf16_huge, the LDS-DMA
    forward / data-gradient kernel with 64-channel chunks, the 400-voxel 3072^2 x 27 level.  Specification in f64 on the GPU."""
    from embodiedscan_amd import engine as E, pipeline
    from embodiedscan_amd.config import build_detector, load_config
    from embodiedscan_amd.synth import make_occ_gt, make_scan
    dev = torch.device('cuda:0')
    cfg = load_config(os.path.join(ROOT, 'configs', 'mv_occ.py'))
    det = build_detector(cfg, device=dev, seed=0).to(dev)
    sc = make_scan(4321, n_views=10, augment=False, render_device=str(dev))
    oc = make_occ_gt(sc, seed=0)
    dscan = pipeline.upload_scan(sc, dev)
    recs, ops = _recorded_step(det, lambda: pipeline.make_occ_batch([dscan], [oc]), lambda: E.TAPE.backward())
    n_dw, n_dx, n_db, kinds = _check_conv_records(recs, 'occupancy (config-5 scale)', dev)
    assert n_dw > 100 and n_dx > 90
    assert any(k[0] == 27 and k[1] == 3072 and k[2] == 3072 for k in kinds) and any(k[0] == 27 and k[1] == 768 for k in kinds)


def test_config4_scale_grounder_step_in_situ():
    """the SHIPPED grounder (20 views 480 x 640, 100 k points, MinkNeck pruning at 1000 voxels, 256 queries, 6 decoder layers, FFN
    2048, RoBERTa-base-shaped text encoder), 2 scans: every backward launch of one bf16 step on its real operands -- the
    decoder's K = 1 256 -> 256 / 256 -> 2048 launches, attention over ~3 000 point tokens, LayerNorm, ContrastiveEmbed,
    plus all convolutions of the backbones / neck.  Specification in f64 on the GPU."""
    from embodiedscan_amd import pipeline
    from embodiedscan_amd.config import build_detector, load_config
    from embodiedscan_amd.synth import make_grounding_sample, make_scan
    dev = torch.device('cuda:0')
    cfg = load_config(os.path.join(ROOT, 'configs', 'mv_grounding.py'))
    det = build_detector(cfg, device=dev, seed=0).to(dev)
    g = torch.Generator().manual_seed(4)                     # non-degenerate regression branch (zero-initialised in the reference)
    sd = {k: v.cpu() for k, v in det.state_dict().items()}
    for k in sd:
        if 'reg_branches' in k and k.endswith('.4.weight'):
            sd[k] = torch.randn(sd[k].shape, generator=g) * 0.05
        if 'reg_branches' in k and k.endswith('.4.bias'):
            sd[k] = torch.randn(sd[k].shape, generator=g) * 0.1
    for k in list(sd):
        if 'reg_branches.' in k and not k.startswith('bbox_head.reg_branches.0.'):
            sd[k] = sd['bbox_head.reg_branches.0.' + k.split('.', 3)[3]]
    det.load_state_dict({k: v.to(dev) for k, v in sd.items()})
    scans = [make_scan(4100 + i, n_views=20, render_device='cuda:0') for i in range(2)]
    anns = [make_grounding_sample(s, seed=40 + i) for i, s in enumerate(scans)]
    dscans = [pipeline.upload_scan(s, dev) for s in scans]
    recs, ops = _recorded_step(det, lambda: pipeline.make_grounding_batch(dscans, anns), lambda: det._backward(None))
    n_dw, n_dx, n_db, kinds = _check_conv_records(recs, 'mv-grounding (config-4 scale)', dev)
    assert n_dw > 150 and n_dx > 150
    assert any(k[0] == 1 and k[1] == 256 and k[2] == 2048 for k in kinds)
    n = _check_ops(ops, 'mv-grounding (config-4 scale)', dev)
    assert n['attn'] == 6 * 3 and n['contrastive'] == 6
