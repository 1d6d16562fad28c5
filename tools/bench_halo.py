"""A/B of the halo-tile K = 27 convolution (csrc/halo.hip) against the gather kernels (spconv.hip) on head-level-like coordinate
sets of synthetic scans (dev tool): time of the plan, of both kernels (forward shape and data-gradient shape), halo statistics,
agreement of the results."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from embodiedscan_amd import engine as E, hip, pipeline, sparse
from embodiedscan_amd.hip import P, call
from embodiedscan_amd.synth import make_scan

dev = torch.device('cuda:0')
nscan = int(sys.argv[1]) if len(sys.argv) > 1 else 4
scans = [make_scan(1234 + i, render_device='cuda:0') for i in range(nscan)]
pts = [pipeline.depth_to_points(pipeline.upload_scan(s, dev)) for s in scans]
cs, _ = sparse.voxelize(pts, 0.01)
s8 = cs.strided(2).strided(2).strided(2)
s16, s32 = s8.strided(2), s8.strided(2).strided(2)
s64 = s32.strided(2)
L2 = s64.children()
L1 = L2.children()
L0 = L1.children()
print('rows: voxels', cs.n, 's8', s8.n, 's16', s16.n, 'L2', L2.n, 'L1', L1.n, 'L0', L0.n)
st = torch.cuda.current_stream().cuda_stream


def timeit(fn, n=10):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


hip.raw('es_halo_set_option')(30, 1)
CASES = ((L0, 128, 128), (L1, 256, 256), (L1, 256, 128), (L1, 128, 256), (s8, 64, 128), (L2, 512, 512), (s8, 128, 128))
if os.environ.get('ONLY_L0'):
    CASES = CASES[:1]
for (S, cin, cout) in CASES:
    n = S.n
    nbr = S.kernel_map(S, 3)
    pairs = int((nbr >= 0).sum())
    x = torch.randn(n, cin, device=dev)
    xh = x.to(torch.bfloat16)
    w = torch.randn(27, cin, cout, device=dev) * 0.05
    wb_n, wb_t = torch.empty((27, cin, cout), dtype=torch.bfloat16, device=dev), torch.empty((27, cout, cin), dtype=torch.bfloat16, device=dev)
    call('es_cast_weight_bf16', P(w), 27, cin, cout, P(wb_n), P(wb_t), st)
    y1, y2 = torch.empty(n, cout, device=dev), torch.empty(n, cout, device=dev)
    ws, nf = E._split_ws(n, 27, cin, cout, x)

    def gather():
        if ws is not None:
            call('es_spconv_fwd_bf16_ws', P(xh), 1, cin, P(wb_t), P(nbr), n, n, 27, cin, cout, 0, P(y1), cout, 0, P(ws), nf, st)
        else:
            call('es_spconv_fwd_bf16', P(xh), 1, cin, P(wb_t), P(nbr), n, n, 27, cin, cout, 0, P(y1), cout, 0, st)

    def plan():
        nbr._halo = None
        E.halo_plan(nbr)
    tp = timeit(plan)
    loc, hrows, hcnt = E.halo_plan(nbr)
    hc = hcnt.float()

    def halo():
        call('es_spconv_halo_bf16', P(xh), cin, P(wb_t), P(loc), P(hrows), P(hcnt), n, n, 27, cin, cout, 0, P(y2), cout, 0, 0, st)
    tg, th = timeit(gather), timeit(halo)
    err = float((y1 - y2).abs().max() / y1.abs().max())
    y3 = torch.empty_like(y2)
    y3.copy_(y2)
    halo()
    same = bool(torch.equal(y2, y3))
    fl = 2.0 * pairs * cin * cout
    print(f'n={n:7d} {cin:4d}->{cout:4d} pairs/row {pairs / n:5.1f} halo mean {float(hc.mean()):6.1f} max {int(hc.max()):4d} | plan {tp * 1e3:7.1f} us | '
          f'gather {tg * 1e3:7.1f} us ({fl / tg / 1e9:6.1f} TF/s) | halo {th * 1e3:7.1f} us ({fl / th / 1e9:6.1f} TF/s) | x{tg / th:4.2f} | '
          f'max rel diff {err:.1e} | rerun identical {same}')
