"""Micro-benchmark of the convolution engine on a head-level-0-like coordinate set (dev tool)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from embodiedscan_amd import sparse, hip
from embodiedscan_amd.hip import P, call
from embodiedscan_amd.engine import _wgrad as WG
from embodiedscan_amd.synth import make_scan
from embodiedscan_amd import pipeline

dev = torch.device('cuda:0')
nscan = int(sys.argv[1]) if len(sys.argv) > 1 else 4
scans = [make_scan(1234 + i, render_device='cuda:0') for i in range(nscan)]
pts = [pipeline.depth_to_points(pipeline.upload_scan(s, dev)) for s in scans]
cs, _ = sparse.voxelize(pts, 0.01)
s8 = cs.strided(2).strided(2).strided(2)
s16 = s8.strided(2)
L0 = s16.children().children()
print('rows: voxels', cs.n, 's8', s8.n, 's16', s16.n, 'L0', L0.n)
st = torch.cuda.current_stream().cuda_stream


def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


cases = ((L0, 128, 128), (s8, 64, 64), (s16.children(), 256, 256))
if os.environ.get('ONLY_L0'):
    cases = cases[:1]
for (S, cin, cout) in cases:
    n = S.n
    nbr = S.kernel_map(S, 3)
    pairs = int((nbr >= 0).sum())
    ident = torch.arange(n, dtype=torch.int32, device=dev)[:, None].repeat(1, 27).contiguous()
    x = torch.randn(n, cin, device=dev)
    w = torch.randn(27, cin, cout, device=dev) * 0.05
    wb_n, wb_t = torch.empty((27, cin, cout), dtype=torch.bfloat16, device=dev), torch.empty((27, cout, cin), dtype=torch.bfloat16, device=dev)
    call('es_cast_weight_bf16', P(w), 27, cin, cout, P(wb_n), P(wb_t), st)
    y = torch.empty(n, cout, device=dev)
    dw = torch.zeros_like(w)
    print(f'--- n={n} {cin}->{cout} pairs={pairs} ({pairs / n:.1f}/row)')
    for tag, m, p_ in (('map', nbr, pairs), ('identity-map(all 27 taps same row)', ident, n * 27)):
        t = timeit(lambda: call('es_spconv_fwd', P(x), cin, P(w), P(m), n, n, 27, cin, cout, 0, P(y), cout, 0, 0, st))
        print(f'  f32 fwd  [{tag}]: {t:.3f} ms  {2 * p_ * cin * cout / t / 1e9:.1f} TF/s alg  gather {p_ * cin * 4 / t / 1e6:.0f} GB/s')
        t = timeit(lambda: call('es_spconv_fwd_bf16', P(x), 0, cin, P(wb_t), P(m), n, n, 27, cin, cout, 0, P(y), cout, 0, st))
        print(f'  bf16 fwd [{tag}]: {t:.3f} ms  {2 * p_ * cin * cout / t / 1e9:.1f} TF/s alg  gather {p_ * cin * 4 / t / 1e6:.0f} GB/s')
        xh = x.to(torch.bfloat16)
        t = timeit(lambda: call('es_spconv_fwd_bf16', P(xh), 1, cin, P(wb_t), P(m), n, n, 27, cin, cout, 0, P(y), cout, 0, st))
        tag = tag + ', bf16 rows'
        print(f'  bf16 fwd [{tag}]: {t:.3f} ms  {2 * p_ * cin * cout / t / 1e9:.1f} TF/s alg  gather {p_ * cin * 4 / t / 1e6:.0f} GB/s')
        t = timeit(lambda: WG('es_spconv_wgrad', st, P(dw), P(x), cin, P(y), cout, P(m), n, n, 27, cin, cout))
        print(f'  f32 wgrad[{tag}]: {t:.3f} ms  {2 * p_ * cin * cout / t / 1e9:.1f} TF/s alg')
        t = timeit(lambda: WG('es_spconv_wgrad_bf16', st, P(dw), P(x), cin, P(y), cout, P(m), n, n, 27, cin, cout))
        print(f'  bf16 wgrad[{tag}]: {t:.3f} ms  {2 * p_ * cin * cout / t / 1e9:.1f} TF/s alg')
    t = timeit(lambda: call('es_spconv_fwd_bf16', P(x), 0, cin, P(wb_t), 0, n, n, 1, cin, cout, 0, P(y), cout, 0, st))
    print(f'  bf16 k1 GEMM: {t:.3f} ms  {2 * n * cin * cout / t / 1e9:.1f} TF/s  read {n * cin * 4 / t / 1e6:.0f} GB/s')
    t = timeit(lambda: y.copy_(x[:, :cout]) if cin >= cout else None)
    print(f'  torch copy n x {cout}: {t:.3f} ms  {2 * n * cout * 4 / t / 1e6:.0f} GB/s')

if os.environ.get('ABLATE'):
    S, cin, cout = cases[0]
    n = S.n
    nbr = S.kernel_map(S, 3)
    x = torch.randn(n, cin, device=dev); y = torch.empty(n, cout, device=dev)
    w = torch.randn(27, cin, cout, device=dev) * 0.05
    wb_n, wb_t = torch.empty((27, cin, cout), dtype=torch.bfloat16, device=dev), torch.empty((27, cout, cin), dtype=torch.bfloat16, device=dev)
    call('es_cast_weight_bf16', P(w), 27, cin, cout, P(wb_n), P(wb_t), st)
    for abl, tag in ((0, 'full'), (1, 'no compute (no LDS reads / MFMA)'), (2, 'no global loads'), (4, 'no LDS stores'), (8, 'no barriers'),
                     (3, 'no compute, no loads'), (6, 'no loads, no stores (compute only)'), (7, 'only barriers+iterator'), (15, 'iterator only'), (9, 'no compute no barriers')):
        t = timeit(lambda: call('es_spconv_fwd_bf16', P(x), 0, cin, P(wb_t), P(nbr), n, n, 27, cin, cout, 0, P(y), cout, abl << 8, st))
        print(f'  ablation {abl:2d} {tag}: {t:.3f} ms')
