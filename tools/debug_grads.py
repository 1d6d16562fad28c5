"""Debug aid (not part of the product path): per-block activation-gradient comparison HIP vs oracle."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from embodiedscan_amd import engine as E, pipeline
from embodiedscan_amd.config import build_detector
from embodiedscan_amd.synth import make_scan
from oracle import model as OM

dev = torch.device('cuda:0')
det = build_detector(os.path.join(ROOT, 'configs/mv_3ddet.py'), device=dev, seed=0).to(dev)
sd = {k: v.cpu() for k, v in det.state_dict().items()}
scans = [make_scan(s, n_views=4, height=240, width=320, img_size=(256, 256), n_points=20000) for s in (11, 12)]
dscans = [pipeline.upload_scan(s, dev) for s in scans]
res = []
for rep in range(2):
    batch = pipeline.make_batch(dscans)
    points_host = [p.cpu() for p in batch['inputs']['points']]
    E.TAPE.clear()
    data = det.data_preprocessor(batch, True)
    det._bind()
    det.arena.grad.zero_()
    det.backbone_3d.trace = []
    E.DEBUG_GRADS = {}
    losses = det.forward(data['inputs'], data['data_samples'], mode='loss')
    E.TAPE.backward()
    torch.cuda.synchronize()
    DBG = E.DEBUG_GRADS
    res.append(([E.DEBUG_GRADS[id(v)].cpu() if id(v) in E.DEBUG_GRADS else None for v in det.backbone_3d.trace],
                [v.d.cpu().clone() for v in det.backbone_3d.trace], det.arena.grad.cpu().clone()))
print('run-to-run max grad diff', float((res[0][2] - res[1][2]).abs().max()), 'max', float(res[0][2].abs().max()))
names = set(det.arena.grad_dict().keys())
osd = {k: v.clone().requires_grad_(k in names) for k, v in sd.items()}
imgs = torch.stack([OM.preprocess_img(torch.from_numpy(s['img']), [123.675, 116.28, 103.53], [58.395, 57.12, 57.375]) for s in scans])
trace = []
ol = OM.detector_loss(osd, points_host, imgs, [s['meta'] for s in scans], [torch.from_numpy(s['gt_boxes']) for s in scans],
                      [torch.from_numpy(s['gt_labels']) for s in scans], trace=trace)
sum(ol.values()).backward()
osd64 = {k: v.double().requires_grad_(k in names) for k, v in sd.items()}
trace64 = []
aux_t = None
ol32, aux32 = OM.detector_loss({k: v.clone() for k, v in sd.items()}, points_host, imgs, [s['meta'] for s in scans],
                               [torch.from_numpy(s['gt_boxes']) for s in scans], [torch.from_numpy(s['gt_labels']) for s in scans], return_aux=True)
l64 = OM.detector_loss(osd64, [p.double() for p in points_host], imgs.double(), [s['meta'] for s in scans],
                       [torch.from_numpy(s['gt_boxes']).double() for s in scans], [torch.from_numpy(s['gt_labels']) for s in scans],
                       targets_override=aux32['targets'], trace=trace64)
sum(l64.values()).backward()
h32 = {n: t for n, t in trace if n.startswith('head')}
h64 = {n: t for n, t in trace64 if n.startswith('head')}
def e3(g, a, b):
    return float((g.double().cpu() - b.grad).abs().max() / b.grad.abs().max()), float((a.grad.double() - b.grad).abs().max() / b.grad.abs().max())
for i, lv in enumerate(det.bbox_head.last_levels):
    dho = lv['dho']
    for nm, sl in (('center', slice(0, 1)), ('reg', slice(1, 13)), ('cls', slice(13, None))):
        eh, eo = e3(dho[:, sl], h32[f'head.L{i}.{nm}'], h64[f'head.L{i}.{nm}'])
        print(f'head.L{i}.{nm}: grad err vs f64: hip {eh:.2e} oracle32 {eo:.2e}')
    g = DBG.get(id(lv['out']))
    if g is not None:
        eh, eo = e3(g, h32[f'head.L{i}.out'], h64[f'head.L{i}.out'])
        print(f'head.L{i}.out: grad err vs f64: hip {eh:.2e} oracle32 {eo:.2e}')
trace = [t for t in trace if not t[0].startswith('head')]
trace64 = [t for t in trace64 if not t[0].startswith('head')]
for (name, t), g, d, (_, t64) in zip(trace, res[0][0], res[0][1], trace64):
    if g is not None and name.endswith('out'):
        eh = float((g.double() - t64.grad).abs().max() / t64.grad.abs().max())
        eo = float((t.grad.double() - t64.grad).abs().max() / t64.grad.abs().max())
        print(f'{name}: grad err vs f64: hip {eh:.2e} oracle32 {eo:.2e}')
for (name, t), g, d in zip(trace, res[0][0], res[0][1]):
    if not name.startswith('layer3.1'):
        continue
    ef = float((d - t.detach()).abs().max() / t.detach().abs().max())
    eg = float((g - t.grad).abs().max() / t.grad.abs().max()) if g is not None else -1
    print(f'{name}: rows {t.shape[0]} feat rel err {ef:.2e} grad rel err {eg:.2e}')

# ---- isolate norm2 of layer3.1: rerun fwd+bwd standalone and against torch f64
tr = det.backbone_3d.trace
names_tr = [n for n, _ in trace]
i_out = names_tr.index('layer3.1.out')
o2, f = tr[i_out - 1], tr[i_out]
f_prev = tr[names_tr.index('layer3.0.out')]
dy = E.DEBUG_GRADS[id(f)].clone()
pre = 'backbone_3d.layer3.1.norm2.bn.'
w, b = det.arena.p[pre + 'weight'], det.arena.p[pre + 'bias']
rec = E.DEBUG_GRADS[('norm', id(f))]
E.TAPE.clear(); E.DEBUG_GRADS = None
xv, rv = E.Var(o2.d.clone()), E.Var(f_prev.d.clone())
wp, bp = E.Param(w.clone(), torch.zeros_like(w)), E.Param(b.clone(), torch.zeros_like(b))
y = E.norm(xv, wp, bp, [0, xv.d.shape[0]], 1e-5, act=1, res=rv)
y.g = dy.clone()
E.TAPE.backward(); torch.cuda.synchronize()
x64 = o2.d.double().cpu().requires_grad_(True); r64 = f_prev.d.double().cpu().requires_grad_(True)
w64 = w.double().cpu().requires_grad_(True); b64 = b.double().cpu().requires_grad_(True)
y64 = torch.relu(torch.nn.functional.batch_norm(x64, None, None, w64, b64, True, 0.1, 1e-5) + r64)
(y64 * dy.double().cpu()).sum().backward()
def rel(a, b): return float((a.double().cpu() - b).abs().max() / b.abs().max())
print('standalone norm: y', rel(y.d, y64.detach()), 'dx', rel(xv.g, x64.grad), 'dres', rel(rv.g, r64.grad), 'dw', rel(wp.g, w64.grad), 'db', rel(bp.g, b64.grad))
ec = (xv.g.double().cpu() - x64.grad).abs().max(0).values
c = int(ec.argmax()); xm = o2.d[:, c].double()
print('worst channel', c, 'err', float(ec[c]), 'mean', float(xm.mean()), 'var', float(xm.var(unbiased=False)), 'w', float(w[c]), 'max|dx|', float(x64.grad.abs().max()))

print('pipeline record: acc', rec['acc'], 'n', rec['n'], 'C', rec['C'], 'act', rec['act'])
print('pipeline dx vs f64', rel(rec['dx'], x64.grad), 'pipeline dz vs standalone dz', float((rec['dz'] - y.g).abs().max()),
      'x same', float((rec['x'] - o2.d).abs().max()), 'yd same', float((rec['yd'] - f.d).abs().max()))
m64 = x64.detach().mean(0); v64 = x64.detach().var(0, unbiased=False)
print('mean err', float((rec['mean'][0].double().cpu() - m64).abs().max()), 'invstd rel err', float(((rec['invstd'][0].double().cpu() - 1 / torch.sqrt(v64 + 1e-5)).abs() * torch.sqrt(v64 + 1e-5)).max()))

# ---- structure of the layer3.1.out gradient error
nm = [n for n, _ in trace]
k = nm.index('layer3.1.out')
g_h = res[0][0][k].double(); g_o = trace[k][1].grad.double(); g_t = trace64[k][1].grad
for tag, g in (('hip', g_h), ('oracle32', g_o)):
    e = (g - g_t)
    print(tag, 'max|g|', float(g_t.abs().max()), 'err max', float(e.abs().max()), 'err fro', float(e.norm()), 'g fro', float(g_t.norm()),
          'top rows', [round(float(v), 9) for v in e.abs().max(1).values.topk(5).values], 'top ch', [round(float(v), 9) for v in e.abs().max(0).values.topk(5).values])
    x = trace64[nm.index('layer3.1.conv2')][1].detach()
    xh = (x - x.mean(0)) / x.std(0, unbiased=False)
    mask = (trace64[k][1].detach() > 0).double()
    ez = e * mask
    proj = ez.mean(0, keepdim=True) + xh * (ez * xh).mean(0, keepdim=True)
    print('   after relu mask: err fro', float(ez.norm()), 'removed by BN-bwd projection', float(proj.norm()), 'residual', float((ez - proj).norm()))
    tz = g_t * mask
    tproj = tz.mean(0, keepdim=True) + xh * (tz * xh).mean(0, keepdim=True)
    print('   true dz fro', float(tz.norm()), 'true residual after projection', float((tz - tproj).norm()))
g_conv_entry = DBG[id(o2)]
print('rec dx vs conv-entry snapshot: max diff', float((rec['dx'] - g_conv_entry).abs().max()), 'max', float(rec['dx'].abs().max()))
t32 = trace[nm.index('layer3.1.conv2')][1].grad; t64 = trace64[nm.index('layer3.1.conv2')][1].grad
print('oracle32 conv2-out grad max', float(t32.abs().max()), 'f64', float(t64.abs().max()), 'rec dx vs t64 rel', float((rec['dx'].double().cpu() - t64).abs().max() / t64.abs().max()),
      'entry vs t64 rel', float((g_conv_entry.double().cpu() - t64).abs().max() / t64.abs().max()))
d = (g_conv_entry.double().cpu() - t64)
print('err top rows', d.abs().max(1).values.topk(5), 'top ch', d.abs().max(0).values.topk(5).indices)
