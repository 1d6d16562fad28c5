"""dev tool: stand-alone timings of the train-mode norm layers (3 + 3 launches each) at the row counts / widths of one mv-3ddet
step, against an elementwise pass over the same bytes -- how much of the 3.7 ms per step is launch latency (small levels) and how
much is bandwidth (the 10^5-row levels).
  python tools/bench_norm.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from embodiedscan_amd import engine as E

dev = torch.device('cuda:0')
E.PRECISION[0] = 'bf16'


def timeit(fn, n=20):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


# (rows, channels, residual) of the 3-D backbone / head levels of 4 synthetic scans
CASES = [(132358, 64, False), (23307, 64, True), (7493, 128, True), (2308, 256, True), (740, 512, True), (5920, 512, False),
         (47360, 256, False), (378880, 128, False), (358208, 128, False)]
CHUNK = int(os.environ.get('NORM_CHUNK', '0'))
if CHUNK:
    from embodiedscan_amd import hip
    hip.raw('es_set_option')(9, CHUNK)
    CASES = CASES[-3:]
for n, C, with_res in CASES:
    x = E.Var(torch.randn(n, C, device=dev))
    w = E.Param(torch.rand(C, device=dev) + 0.5, torch.zeros(C, device=dev))
    b = E.Param(torch.randn(C, device=dev), torch.zeros(C, device=dev))
    res = E.Var(torch.randn(n, C, device=dev)) if with_res else None
    dy = torch.randn(n, C, device=dev)

    def fwd():
        E.TAPE.clear()
        return E.norm(x, w, b, [0, n], 1e-5, act=1, res=res)

    def fwd_bwd():
        x.g = None
        if res is not None:
            res.g = None
        y = fwd()
        y.g = dy.clone()
        E.TAPE.backward()
    t_f, t_fb = timeit(fwd), timeit(fwd_bwd)
    a, o = torch.randn(n, C, device=dev), torch.empty(n, C, device=dev)
    t_ref = timeit(lambda: torch.add(a, dy, out=o))
    print(f'rows {n:7d} x {C:4d} res {int(with_res)}: forward {t_f:7.1f} us, forward + backward {t_fb:7.1f} us   | one elementwise pass '
          f'(2 reads + 1 write, {3 * n * C * 4 / 1e6:6.1f} MB) {t_ref:6.1f} us', flush=True)
