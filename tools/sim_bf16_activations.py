"""What would bf16 storage of activations cost in parity?  Simulated in the CPU oracle at config-1 scale:
  A: f32 everywhere (oracle)
  B: GEMM operands rounded to bf16 (what the HIP path does today)
  C: B + every conv / norm / pooling output stored in bf16 (round-3 plan)"""
import sys, torch, numpy as np
sys.path.insert(0, '/root/repo')
from oracle import sparse as S, model as OM, pipeline as OP
from embodiedscan_amd.synth import make_scan
from embodiedscan_amd.params import ParamArena, detector_specs

torch.manual_seed(0)
arena = ParamArena(detector_specs(284), seed=0)
sd = arena.state_dict()
scans = [make_scan(s, n_views=4, height=240, width=320, img_size=(256, 256), n_points=20000) for s in (11, 12)]
pts = [OP.scan_to_points(s) for s in scans]
imgs = torch.stack([OM.preprocess_img(torch.from_numpy(s['img']), [123.675, 116.28, 103.53], [58.395, 57.12, 57.375]) for s in scans])
metas = [s['meta'] for s in scans]
gb = [torch.from_numpy(s['gt_boxes']) for s in scans]; gl = [torch.from_numpy(s['gt_labels']) for s in scans]
r = lambda t: t.bfloat16().float()
orig = dict(gather_conv=S.gather_conv, instance_norm=S.instance_norm, batch_norm=S.batch_norm, max_pool=S.max_pool,
            conv2d=torch.nn.functional.conv2d)

def run(mode):
    def gconv(feats, nbr, weight):
        if mode in 'BC':
            y = orig['gather_conv'](r(feats), nbr, r(weight))
        else:
            y = orig['gather_conv'](feats, nbr, weight)
        return r(y) if mode == 'C' else y
    S.gather_conv = gconv
    if mode == 'C':
        S.instance_norm = lambda *a, **k: (lambda o: o.new(r(o.feats)) if hasattr(o, 'feats') else r(o))(orig['instance_norm'](*a, **k))
        S.batch_norm = lambda *a, **k: (lambda o: o.new(r(o.feats)) if hasattr(o, 'feats') else r(o))(orig['batch_norm'](*a, **k))
    else:
        S.instance_norm, S.batch_norm = orig['instance_norm'], orig['batch_norm']
    def c2d(x, w, *a, **k):
        if mode in 'BC':
            y = orig['conv2d'](r(x), r(w), *a, **k)
        else:
            y = orig['conv2d'](x, w, *a, **k)
        return r(y) if mode == 'C' else y
    torch.nn.functional.conv2d = c2d
    try:
        out = OM.detector_loss(sd, pts, imgs, metas, gb, gl, return_aux=True)
    finally:
        S.gather_conv, S.instance_norm, S.batch_norm = orig['gather_conv'], orig['instance_norm'], orig['batch_norm']
        torch.nn.functional.conv2d = orig['conv2d']
    return out

res = {}
for m in 'ABC':
    try:
        o = run(m)
        losses = o[0] if isinstance(o, tuple) else o
        res[m] = {k: float(v) for k, v in losses.items()}
        print(m, res[m])
    except Exception as e:
        import traceback; traceback.print_exc(); print(m, 'failed', e)
for m in 'BC':
    if m in res:
        print(m, 'rel err vs A:', {k: abs(res[m][k] - res['A'][k]) / abs(res['A'][k]) for k in res['A']})
