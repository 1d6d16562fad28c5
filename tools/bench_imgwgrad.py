"""A/B of the 3x3 image weight-gradient kernel (csrc/imgwgrad.hip) against the map kernel on the image backbone's shapes (dev tool)"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from embodiedscan_amd.engine import _wgrad as WG
from embodiedscan_amd.hip import P, call, raw

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


for n_img, H, W, C, S in ((80, 60, 60, 32, 1), (80, 30, 30, 64, 1), (240, 60, 60, 32, 1), (240, 30, 30, 64, 1), (80, 120, 120, 32, 1),
                          (80, 120, 120, 32, 2), (80, 60, 60, 64, 2)):      # mv-3ddet (4 scans x 20 views), grounding (12 x 20), a larger map, the stride-2 layers
    n, n_o = n_img * H * W, n_img * (H // S) * (W // S)
    xh = torch.randn(n, C, device=dev).to(torch.bfloat16)
    gy = torch.randn(n_o, C, device=dev)
    nbr = torch.empty((n_o, 9), dtype=torch.int32, device=dev)
    call('es_image_map', n_img, H, W, H // S, W // S, 3, 3, S, 1, P(nbr), st)
    d1, d2 = torch.zeros(9, C, C, device=dev), torch.zeros(9, C, C, device=dev)
    t_map = timeit(lambda: WG('es_spconv_wgrad_bf16_src', st, P(d1), P(xh), 1, C, P(gy), 0, C, P(nbr), n_o, n, 9, C, C))
    d1.zero_()
    WG('es_spconv_wgrad_bf16_src', st, P(d1), P(xh), 1, C, P(gy), 0, C, P(nbr), n_o, n, 9, C, C)      # (accumulates after its first call of an epoch)
    res = []
    for wgs in ((100, 200, 400, 800) if C == 32 else (40, 80, 160, 320)):
        raw('es_img_wgrad_set_option')(41 if C == 32 else 42, wgs)
        nf = int(raw('es_img_wgrad9_workspace_floats')(n_img, H, W, C, S))
        ws = torch.empty(nf, device=dev)
        t = timeit(lambda: call('es_img_wgrad9_bf16', P(xh), C, P(gy), C, n_img, H, W, C, S, P(d2), 0, P(ws), nf, st))
        res.append(f'{wgs} wgs {t:6.1f} us')
    mb = (n * 2 + n_o * 4) * C / 1e6
    print(f'{n_img} x {H} x {W} x {C} stride {S}: {mb:.0f} MB of operands | map kernel {t_map:6.1f} us | image kernel ' + ', '.join(res) +
          f' | max rel diff {float((d1 - d2).abs().max() / d1.abs().max()):.1e}')
raw('es_img_wgrad_set_option')(41, 400)
raw('es_img_wgrad_set_option')(42, 160)
print('--- 1x1 layers on contiguous rows: map kernel vs streaming kernel')
raw('es_img_wgrad_set_option')(43, 0)
for n, cin, cout in ((288000, 32, 128), (288000, 128, 32), (72000, 64, 256), (72000, 256, 64), (18000, 128, 512), (18000, 512, 128), (1152000, 64, 32),
                     (864000, 32, 128), (864000, 128, 32), (216000, 64, 256), (216000, 256, 64), (352224, 128, 320), (60224, 128, 320), (7528, 128, 320)):
    xh = torch.randn(n, cin, device=dev).to(torch.bfloat16)
    gy = torch.randn(n, cout, device=dev)
    d1, d2 = torch.zeros(1, cin, cout, device=dev), torch.zeros(cin, cout, device=dev)
    t1 = timeit(lambda: WG('es_spconv_wgrad_bf16_src', st, P(d1), P(xh), 1, cin, P(gy), 0, cout, 0, n, n, 1, cin, cout))
    nf = int(raw('es_rows_wgrad1_workspace_floats')(n, cin, cout))
    ws = torch.empty(nf, device=dev)
    t2 = timeit(lambda: call('es_rows_wgrad1_bf16', P(xh), cin, P(gy), cout, n, cin, cout, P(d2), 0, P(ws), nf, st))
    mb = n * (cin * 2 + cout * 4) / 1e6
    print(f'{n} x {cin}->{cout}: {mb:.0f} MB | map kernel {t1:6.1f} us ({mb / t1 / 1e3 * 1e3:.2f} TB/s) | rows kernel {t2:6.1f} us ({mb / t2:.2f} TB/s), {nf // (cin * cout)} slices')

print('--- the head\'s 128 -> 320 forward GEMM (K = 1): 64-column tiles vs one 320-column tile (option 23)')
for n in (352224, 60224, 7528):
    cin, cout = 128, 320
    x = torch.randn(n, cin, device=dev)
    w = torch.randn(1, cin, cout, device=dev) * 0.05
    bias = torch.randn(cout, device=dev)
    wn, wt = torch.empty((1, cin, cout), dtype=torch.bfloat16, device=dev), torch.empty((1, cout, cin), dtype=torch.bfloat16, device=dev)
    call('es_cast_weight_bf16', P(w), 1, cin, cout, P(wn), P(wt), st)
    ys, ts = [], []
    for on in (0, 1):
        raw('es_set_option')(23, on)
        y = torch.empty(n, cout, device=dev)
        ts.append(timeit(lambda: call('es_spconv_fwd_bf16', P(x), 0, cin, P(wt), 0, n, n, 1, cin, cout, P(bias), P(y), cout, 0, st)))
        ys.append(y)
    mb = n * (cin * 4 + cout * 4) / 1e6
    print(f'{n} x {cin}->{cout}: {mb:.0f} MB | 64-column tiles {ts[0]:6.1f} us ({mb / ts[0]:.2f} TB/s) | one tile {ts[1]:6.1f} us ({mb / ts[1]:.2f} TB/s) | bit-identical {bool(torch.equal(ys[0], ys[1]))}')
