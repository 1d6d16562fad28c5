"""dev tool: stand-alone timings of the K = 1 row-GEMM launches the image backbone spends its time in (shapes of one mv-3ddet
step at 80 images of 480 x 480: expansion 1x1 convolutions with bf16 rows in / out and a bf16 residual, reduction 1x1
convolutions, the head's 128 -> 320 output GEMM) against a plain elementwise pass over the same bytes (torch.add on bf16 /
f32 tensors) -- how far each launch is from what the memory system gives a streaming kernel of the same footprint.
  python tools/bench_rowgemm.py [--opt key=value,...]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from embodiedscan_amd import hip
from embodiedscan_amd.hip import P, call

ap = argparse.ArgumentParser()
ap.add_argument('--opt', default='')
ap.add_argument('--reps', type=int, default=20)
ap.add_argument('--abl', type=int, default=0, help='ablation bits of k_rowgemm2_bf16 (1 no stores, 2 no input loads, 4 no residual loads)')
ap.add_argument('--first', type=int, default=99)
args = ap.parse_args()
for kv in [p for p in args.opt.split(',') if p]:
    k, v = kv.split('=')
    hip.raw('es_set_option')(int(k), int(v))
dev = torch.device('cuda:0')
st = torch.cuda.current_stream().cuda_stream


def timeit(fn, n=args.reps):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3            # us


g = torch.Generator().manual_seed(0)
h16 = torch.bfloat16
# (rows, cin, cout, x bf16, y bf16, residual: None / 'bf16' / 'f32', act)
CASES = [(1152000, 16, 64, 1, 1, 'bf16', 1), (1152000, 16, 64, 1, 1, None, 0), (288000, 32, 128, 1, 1, 'bf16', 1),
         (72000, 64, 256, 1, 1, 'bf16', 1), (18000, 128, 512, 1, 1, 'bf16', 1), (1152000, 64, 16, 1, 1, None, 1),
         (288000, 128, 32, 1, 1, None, 1), (72000, 256, 64, 1, 1, None, 1), (358208, 128, 320, 0, 0, None, 0),
         (358208, 320, 128, 0, 0, None, 0), (288000, 32, 128, 0, 0, None, 0), (288000, 128, 32, 0, 0, 'f32', 3)]
for n, cin, cout, xh, yh, res, act in CASES[:args.first]:
    x = torch.randn(n, cin, generator=g).to(dev)
    xs = x.to(h16).contiguous() if xh else x
    w = (torch.randn(1, cin, cout, generator=g) / cin ** 0.5).to(dev)
    wt = torch.empty((1, cout, cin), dtype=h16, device=dev)
    wn = torch.empty((1, cin, cout), dtype=h16, device=dev)
    call('es_cast_weight_bf16', P(w), 1, cin, cout, P(wn), P(wt), st)
    scale, shift = (torch.rand(cout, generator=g) + 0.5).to(dev), torch.randn(cout, generator=g).to(dev)
    r = None
    if res is not None:
        r = torch.randn(n, cout, generator=g).to(dev)
        r = r.to(h16).contiguous() if res == 'bf16' else r
    y = torch.empty(n, cout, device=dev, dtype=h16 if yh else torch.float32)
    nbytes = xs.numel() * xs.element_size() + y.numel() * y.element_size() + (r.numel() * r.element_size() if r is not None else 0)

    def launch():
        call('es_spconv_fwd_bf16_io', P(xs), xh, cin, P(wt), 0, n, n, 1, cin, cout, P(scale), P(shift) if act != 3 else 0,
             P(r) if r is not None else 0, int(res == 'bf16'), cout if r is not None else 0, act, P(y), (yh | (args.abl << 8)) if yh else 0, cout, st)
    t = timeit(launch)
    # the same bytes through an elementwise pass: out = a (+ b), with `a` as wide as the output and an extra read of the input
    a = torch.empty_like(y)
    b = r if r is not None else None
    t_ref = timeit((lambda: torch.add(a, b, out=y)) if b is not None else (lambda: y.copy_(a)))
    ref_bytes = y.numel() * y.element_size() * (3 if b is not None else 2)
    print(f'n={n:8d} {cin:4d}->{cout:4d} x:{"bf16" if xh else "f32 "} y:{"bf16" if yh else "f32 "} res:{str(res):5s} act {act}: '
          f'{t:7.1f} us  {nbytes / t / 1e3:7.1f} GB/s   | elementwise pass over {ref_bytes / 1e6:6.1f} MB: {t_ref:7.1f} us '
          f'{ref_bytes / t_ref / 1e3:7.1f} GB/s', flush=True)
