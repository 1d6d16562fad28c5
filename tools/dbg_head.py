import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from embodiedscan_amd import engine as E, pipeline
from embodiedscan_amd.config import build_detector, load_config
from embodiedscan_amd.synth import make_scan
dev = torch.device('cuda:0')
det = build_detector(load_config(os.path.join(ROOT, 'configs', 'mv_3ddet.py')), device=dev, seed=0).to(dev)
dscans = [pipeline.upload_scan(make_scan(1234 + i, n_views=4, render_device='cuda:0'), dev) for i in range(2)]
batch = pipeline.make_batch(dscans)
rel = lambda a, b: float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))
for mode in ('f32', 'bf16'):
    E.PRECISION[0] = mode
    E.TAPE.clear(); E.WEIGHT_VERSION[0] += 1
    data = det.data_preprocessor(batch, True)
    det._bind(); det.arena.grad.zero_()
    det.forward(data['inputs'], data['data_samples'], mode='loss')
    lv = det.bbox_head.last_levels[0]
    gy = lv['ho'].g.clone()
    W = det.bbox_head.head_w.d[0]                       # (128, 320)
    ref = gy @ W.t()
    blocks = [(0, 1), (1, 13), (13, 297), (297, 320)]
    print(mode, 'gy col-block norms', [f'{float(gy[:, a:b].norm()):.3e}' for a, b in blocks], 'W col-block norms', [f'{float(W[:, a:b].norm()):.3e}' for a, b in blocks])
    E.DEBUG_GRADS = {}
    E.TAPE.backward(); torch.cuda.synchronize()
    got = lv['out'].g if lv['out'].g is not None else None
    dg = E.DEBUG_GRADS.get(id(lv['out']))
    print(mode, 'head dgrad: |ref|', float(ref.norm()), 'recorded', None if dg is None else (float(dg.norm()), rel(dg, ref)))
    if mode == 'bf16':
        bn, bt = det.bbox_head.head_w.bf16()
        print('bf16 copies: natural', rel(bn[0].float(), W), 'transposed', rel(bt[0].float(), W.t()))
    E.DEBUG_GRADS = None
