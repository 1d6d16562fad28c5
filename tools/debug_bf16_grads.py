"""dev tool: walk the backward pass of one train step in f32 and bf16 mode and compare, conv by conv (in backward
order), the output gradient each convolution receives -- finds the first op whose gradient deviates."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from embodiedscan_amd import engine as E, pipeline
from embodiedscan_amd.config import build_detector, load_config
from embodiedscan_amd.synth import make_scan
if os.environ.get('NO_GATE'):
    _ca = E.conv_affine
    E.conv_affine = lambda *a, **k: _ca(*a, **{**k, 'sole_consumer': False})
dev = torch.device('cuda:0')
cfg = load_config(os.path.join(ROOT, 'configs', 'mv_3ddet.py'))
det = build_detector(cfg, device=dev, seed=0).to(dev)
nv = int(os.environ.get('VIEWS', 4))
dscans = [pipeline.upload_scan(make_scan(1234 + i, n_views=nv, render_device='cuda:0'), dev) for i in range(2)]
batch = pipeline.make_batch(dscans)


class Rec(dict):
    def __init__(self):
        super().__init__(); self.seq = []
    def __setitem__(self, k, v):
        self.seq.append(v)


recs = {}
for mode in ('f32', 'bf16'):
    E.PRECISION[0] = mode
    E.TAPE.clear(); E.WEIGHT_VERSION[0] += 1
    data = det.data_preprocessor(batch, True)
    det._bind(); det.arena.grad.zero_()
    losses = det.forward(data['inputs'], data['data_samples'], mode='loss')
    if mode == 'f32':
        seeds = [lv['ho'].g.clone() for lv in det.bbox_head.last_levels]
    elif os.environ.get('SAME_SEED', '1') == '1':     # same loss gradient in both modes: isolates the backward kernels
        for lv, sd in zip(det.bbox_head.last_levels, seeds):
            lv['ho'].g.copy_(sd)
    pg = None
    E.DEBUG_GRADS = Rec()
    E.TAPE.backward(); torch.cuda.synchronize()
    recs[mode] = E.DEBUG_GRADS.seq
    recs[mode + '_pg'] = {k: v.clone() for k, v in det.arena.grad_dict().items()}
    E.DEBUG_GRADS = None
E.PRECISION[0] = 'f32'
a, b = recs['f32'], recs['bf16']
print('ops recorded', len(a), len(b))
for i, (x, y) in enumerate(zip(a, b)):
    tag = 'dY'
    if isinstance(x, dict):
        x, y, tag = x['dz'], y['dz'], 'norm dz'
    if x.shape != y.shape:
        print(i, 'shape mismatch', tuple(x.shape), tuple(y.shape)); continue
    n = float(x.norm())
    r = float((x - y).norm()) / n if n > 0 else 0.0
    print(f'{i:3d} {tag} {tuple(x.shape)} |f32|={n:.3e} |bf16|={float(y.norm()):.3e} rel={r:.3e}')
    if i < 0:
        for c0 in range(0, x.shape[1], 27):
            xs, ys = x[:, c0:c0 + 27], y[:, c0:c0 + 27]
            print(f'     cols {c0}-{c0 + 26}: |f32|={float(xs.norm()):.3e} |bf16|={float(ys.norm()):.3e} rel={float((xs - ys).norm() / xs.norm()):.3e}')
        for r0 in range(0, x.shape[0], 4096):
            xs, ys = x[r0:r0 + 4096], y[r0:r0 + 4096]
            print(f'     rows {r0}: |f32|={float(xs.norm()):.3e} |bf16|={float(ys.norm()):.3e} rel={float((xs - ys).norm() / xs.norm()):.3e}')

ga, gb = recs['f32_pg'], recs['bf16_pg']
for k, g in ga.items():
    n = float(g.norm())
    if n > 1e-9:
        print(f'{float((gb[k] - g).norm()) / n:9.3e} |g32|={n:9.3e} {k}')
