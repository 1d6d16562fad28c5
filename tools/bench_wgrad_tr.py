"""A/B of the 128 x 128 transposed-read weight-gradient tile at 32 / 64 pairs per chunk (es_set_option 14 = 1 / 2) on head-like sets (dev tool)"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from embodiedscan_amd import pipeline, sparse
from embodiedscan_amd.engine import _wgrad as WG
from embodiedscan_amd.hip import P, raw
from embodiedscan_amd.synth import make_scan

dev = torch.device('cuda:0')
scans = [make_scan(1234 + i, render_device='cuda:0') for i in range(4)]
pts = [pipeline.depth_to_points(pipeline.upload_scan(s, dev)) for s in scans]
cs, _ = sparse.voxelize(pts, 0.01)
s8 = cs.strided(2).strided(2).strided(2)
s16, s32 = s8.strided(2), s8.strided(2).strided(2)
s64 = s32.strided(2)
L2 = s64.children(); L1 = L2.children(); L0 = L1.children()
st = torch.cuda.current_stream().cuda_stream


def timeit(fn, n=6):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for S, cin, cout in ((L0, 128, 128), (L1, 256, 256), (L1, 256, 128), (L2, 512, 512), (s16, 128, 128), (s32, 256, 256), (s64, 512, 512)):
    n = S.n
    nbr = S.kernel_map(S, 3)
    xh = torch.randn(n, cin, device=dev).to(torch.bfloat16)
    gh = torch.randn(n, cout, device=dev).to(torch.bfloat16)
    out = []
    for opt in (1, 2):
        raw('es_set_option')(14, opt)
        dw = torch.zeros(27, cin, cout, device=dev)
        t = timeit(lambda: WG('es_spconv_wgrad_bf16_src', st, P(dw), P(xh), 1, cin, P(gh), 1, cout, P(nbr), n, n, 27, cin, cout))
        out.append(t)
    print(f'n={n:7d} {cin}->{cout}: 32 pairs / chunk {out[0]:7.1f} us | 64 pairs / chunk {out[1]:7.1f} us | x{out[0] / out[1]:.2f}')
raw('es_set_option')(14, 1)
