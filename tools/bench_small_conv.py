"""dev tool: the K = 27 forward launches of the 3-D backbone's SMALL levels (a few thousand rows, 128 - 512 channels: tap-split
launches of a few hundred workgroups that mostly re-read their weight slices), option by option.
  python tools/bench_small_conv.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from embodiedscan_amd import sparse, hip, pipeline
from embodiedscan_amd.hip import P, call
from embodiedscan_amd.synth import make_scan

dev = torch.device('cuda:0')
scans = [make_scan(1234 + i, render_device='cuda:0') for i in range(4)]
pts = [pipeline.depth_to_points(pipeline.upload_scan(s, dev)) for s in scans]
cs, _ = sparse.voxelize(pts, 0.01)
st = torch.cuda.current_stream().cuda_stream


def timeit(fn, n=20):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


levels, S = [], cs
for c in (64, 64, 128, 256, 512):
    levels.append((S, c))
    S = S.strided(2)
OPTS = [('default', ()), ('weight-sharing order (20=1)', ((20, 1, 0),))]
for S, c in levels[1:]:
    n = S.n
    nbr = S.kernel_map(S, 3)
    pairs = int((nbr >= 0).sum())
    x = torch.randn(n, c, device=dev).to(torch.bfloat16)
    w = torch.randn(27, c, c, device=dev) * 0.05
    wb_n, wb_t = torch.empty((27, c, c), dtype=torch.bfloat16, device=dev), torch.empty((27, c, c), dtype=torch.bfloat16, device=dev)
    call('es_cast_weight_bf16', P(w), 27, c, c, P(wb_n), P(wb_t), st)
    nf = int(hip.raw('es_spconv_split_workspace_floats')(n, 27, c, c))
    ws = torch.zeros(max(nf, 1), device=dev)
    print(f'--- rows {n}  {c} -> {c}  pairs/row {pairs / n:.1f}  split workspace {nf * 4 / 1e6:.1f} MB   weights {27 * c * c * 2 / 1e6:.1f} MB', flush=True)
    ref = None
    for tag, opts in OPTS:
        for k, v, _ in opts:
            hip.raw('es_set_option')(k, v)
        y = torch.empty(n, c, device=dev)
        t = timeit(lambda: call('es_spconv_fwd_bf16_ws', P(x), 1, c, P(wb_t), P(nbr), n, n, 27, c, c, 0, P(y), c, 0, P(ws), nf, st))
        for k, _, v0 in opts:
            hip.raw('es_set_option')(k, v0)
        same = '' if ref is None else f'  bit-identical {bool(torch.equal(y, ref))}'
        ref = y if ref is None else ref
        print(f'  {tag:32s} {t:7.1f} us  {2 * pairs * c * c / t / 1e6:6.1f} TF/s{same}', flush=True)
