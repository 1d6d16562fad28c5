"""A/B of the 3x3 image convolution kernels (csrc/imgconv.hip) against the map kernels on the image backbone's shapes (dev tool)"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from embodiedscan_amd.hip import P, call

dev = torch.device('cuda:0')
st = torch.cuda.current_stream().cuda_stream


def timeit(fn, n=10):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for n_img, H, W, C, S in ((80, 120, 120, 16, 1), (80, 60, 60, 32, 1), (80, 30, 30, 64, 1), (80, 120, 120, 32, 2), (240, 60, 60, 32, 1)):
    n, n_o = n_img * H * W, n_img * (H // S) * (W // S)
    xh = torch.randn(n, C, device=dev).to(torch.bfloat16)
    w = torch.randn(9, C, C, device=dev) * 0.05
    wn, wt = torch.empty((9, C, C), dtype=torch.bfloat16, device=dev), torch.empty((9, C, C), dtype=torch.bfloat16, device=dev)
    call('es_cast_weight_bf16', P(w), 9, C, C, P(wn), P(wt), st)
    scale, shift = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
    nbr = torch.empty((n_o, 9), dtype=torch.int32, device=dev)
    call('es_image_map', n_img, H, W, H // S, W // S, 3, 3, S, 1, P(nbr), st)
    y = torch.empty((n_o, C), dtype=torch.bfloat16, device=dev)
    t_map = timeit(lambda: call('es_spconv_fwd_bf16_io', P(xh), 1, C, P(wt), P(nbr), n_o, n, 9, C, C, P(scale), P(shift), 0, 0, 0, 1, P(y), 1, C, st))
    res = []
    for wgs in (512, 1024, 2048):
        call('es_img_conv_set_option', 51, wgs)
        res.append(f'{wgs}: {timeit(lambda: call("es_img_conv3_bf16", P(xh), C, P(wt), n_img, H, W, C, S, 0, P(scale), P(shift), 0, 0, 1, P(y), 1, C, st)):6.1f}')
    line = f'{n_img} x {H} x {W} x {C} stride {S}: forward map kernel {t_map:6.1f} us | image kernel ' + ', '.join(res)
    if S == 1 and C >= 32:
        inv = torch.empty((n, 9), dtype=torch.int32, device=dev)
        call('es_inverse_map', P(nbr), n, 9, n, P(inv), st)
        gy, dx = torch.randn(n, C, device=dev), torch.empty(n, C, device=dev)
        t1 = timeit(lambda: call('es_spconv_fwd_bf16_io', P(gy), 0, C, P(wn), P(inv), n, n, 9, C, C, P(scale), 0, P(xh), 1, C, 3, P(dx), 0, C, st))
        call('es_img_conv_set_option', 51, 1024)
        t2 = timeit(lambda: call('es_img_conv3_bf16', P(gy), C, P(wn), n_img, H, W, C, 1, 1, P(scale), 0, P(xh), C, 3, P(dx), 0, C, st))
        line += f' || gated dgrad map {t1:6.1f} us | image {t2:6.1f} us'
    print(line)
call('es_img_conv_set_option', 51, 1024)
