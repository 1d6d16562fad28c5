"""3x3 image convolution kernels (csrc/imgconv.hip) on the GPU through the C ABI: forward (stride 1 / 2) and gated data gradient against
the map-kernel path they replace (es_spconv_fwd_bf16_io on es_image_map / es_inverse_map: the oracle-pinned path of
tests/test_gpu_resnet2d.py) on the same operands at the backbone's real shapes, run-to-run bit-identical."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(n_img, H, W, C, S, seed=3):
    from embodiedscan_amd.hip import P, call
    dev = torch.device('cuda:0')
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(seed)
    n, n_o = n_img * H * W, n_img * (H // S) * (W // S)
    x = torch.randn(n, C, generator=g).to(dev)
    w = (torch.randn(9, C, C, generator=g) / (9 * C) ** 0.5).to(dev)
    scale, shift = (0.5 + torch.rand(C, generator=g)).to(dev), torch.randn(C, generator=g).to(dev)
    wn, wt = torch.empty((9, C, C), dtype=torch.bfloat16, device=dev), torch.empty((9, C, C), dtype=torch.bfloat16, device=dev)
    call('es_cast_weight_bf16', P(w), 9, C, C, P(wn), P(wt), st)
    nbr = torch.empty((n_o, 9), dtype=torch.int32, device=dev)
    call('es_image_map', n_img, H, W, H // S, W // S, 3, 3, S, 1, P(nbr), st)
    return dev, st, n, n_o, x, wn, wt, scale, shift, nbr


@pytest.mark.parametrize('case', [(20, 120, 120, 16, 1), (20, 60, 60, 32, 1), (20, 30, 30, 64, 1), (20, 120, 120, 32, 2),
                                  (3, 17, 45, 32, 1), (2, 10, 22, 32, 2)])
def test_forward_vs_map_kernel(case):
    from embodiedscan_amd.hip import P, call, raw
    n_img, H, W, C, S = case
    dev, st, n, n_o, x, wn, wt, scale, shift, nbr = _setup(*case)
    assert raw('es_img_conv3_supported')(n_img, H, W, C, S, 0) == 1
    xh = x.to(torch.bfloat16)
    y1 = torch.zeros((n_o, C), dtype=torch.bfloat16, device=dev)
    call('es_spconv_fwd_bf16_io', P(xh), 1, C, P(wt), P(nbr), n_o, n, 9, C, C, P(scale), P(shift), 0, 0, 0, 1, P(y1), 1, C, st)
    y2 = torch.full((n_o, C), float('nan'), dtype=torch.bfloat16, device=dev)
    call('es_img_conv3_bf16', P(xh), C, P(wt), n_img, H, W, C, S, 0, P(scale), P(shift), 0, 0, 1, P(y2), 1, C, st)
    y3 = torch.full((n_o, C), float('nan'), dtype=torch.bfloat16, device=dev)
    call('es_img_conv3_bf16', P(xh), C, P(wt), n_img, H, W, C, S, 0, P(scale), P(shift), 0, 0, 1, P(y3), 1, C, st)
    yf = torch.full((n_o, C), float('nan'), device=dev)
    call('es_img_conv3_bf16', P(xh), C, P(wt), n_img, H, W, C, S, 0, P(scale), P(shift), 0, 0, 1, P(yf), 0, C, st)
    f1 = torch.zeros((n_o, C), device=dev)
    call('es_spconv_fwd_bf16_io', P(xh), 1, C, P(wt), P(nbr), n_o, n, 9, C, C, P(scale), P(shift), 0, 0, 0, 1, P(f1), 0, C, st)
    torch.cuda.synchronize()
    assert torch.equal(y2.view(torch.int16), y3.view(torch.int16)), 'two runs differ'
    assert torch.equal(yf.to(torch.bfloat16).view(torch.int16), y2.view(torch.int16))      # the bf16 rows are the f32 result rounded
    err = float((yf - f1).abs().max() / f1.abs().max())
    ulp = (y1.view(torch.int16).int() - y2.view(torch.int16).int()).abs()
    print(f'{case}: f32 rows vs map kernel {err:.2e} (tol 3e-6); bf16 rows differing by one ulp: {float((ulp > 0).float().mean()):.2e} (max {int(ulp.max())})')
    assert err < 3e-6 and int(ulp.max()) <= 1


@pytest.mark.parametrize('case', [(20, 60, 60, 32), (20, 30, 30, 64), (3, 17, 45, 32)])
def test_gated_data_gradient_vs_map_kernel(case):
    from embodiedscan_amd.hip import P, call, raw
    n_img, H, W, C = case
    dev, st, n, n_o, x, wn, wt, scale, shift, nbr = _setup(n_img, H, W, C, 1, seed=8)
    assert raw('es_img_conv3_supported')(n_img, H, W, C, 1, 1) == 1
    inv = torch.empty((n, 9), dtype=torch.int32, device=dev)
    call('es_inverse_map', P(nbr), n, 9, n, P(inv), st)
    gy = torch.randn(n, C, device=dev)
    act = x.to(torch.bfloat16)
    d1 = torch.zeros((n, C), device=dev)
    call('es_spconv_fwd_bf16_io', P(gy), 0, C, P(wn), P(inv), n, n, 9, C, C, P(scale), 0, P(act), 1, C, 3, P(d1), 0, C, st)
    d2 = torch.full((n, C), float('nan'), device=dev)
    call('es_img_conv3_bf16', P(gy), C, P(wn), n_img, H, W, C, 1, 1, P(scale), 0, P(act), C, 3, P(d2), 0, C, st)
    d3 = torch.full((n, C), float('nan'), device=dev)
    call('es_img_conv3_bf16', P(gy), C, P(wn), n_img, H, W, C, 1, 1, P(scale), 0, P(act), C, 3, P(d3), 0, C, st)
    torch.cuda.synchronize()
    assert torch.equal(d2, d3)
    err = float((d1 - d2).abs().max() / d1.abs().max())
    print(f'{case}: gated data gradient vs map kernel {err:.2e} (tol 3e-6); zero rows agree: {bool(((d1 == 0) == (d2 == 0)).all())}')
    assert err < 3e-6 and bool(((d1 == 0) == (d2 == 0)).all())


@pytest.mark.parametrize('case', [(20, 480, 640, 16), (3, 97, 131, 16), (2, 64, 64, 32), (10, 480, 640, 64)])
def test_stem_and_max_pool_in_one_launch(case):
    """es_stem_pool_fwd against es_stem_conv_fwd + es_maxpool_fwd_h (the oracle-pinned pair of tests/test_gpu_resnet2d.py): same bits"""
    from embodiedscan_amd.hip import P, call
    n_img, H, W, C = case
    dev = torch.device('cuda:0')
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(4)
    x = torch.randn(n_img, H, W, 3, generator=g).to(dev)
    w = (torch.randn(49, 3, C, generator=g) / 12).to(dev)
    scale, shift = (0.5 + torch.rand(C, generator=g)).to(dev), torch.randn(C, generator=g).to(dev)
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    Hp, Wp = (Ho - 1) // 2 + 1, (Wo - 1) // 2 + 1
    y = torch.empty((n_img * Ho * Wo, C), device=dev)
    nbr = torch.empty((n_img * Hp * Wp, 9), dtype=torch.int32, device=dev)
    call('es_image_map', n_img, Ho, Wo, Hp, Wp, 3, 3, 2, 1, P(nbr), st)
    want = torch.empty((n_img * Hp * Wp, C), dtype=torch.bfloat16, device=dev)
    got = torch.full((n_img * Hp * Wp, C), float('nan'), dtype=torch.bfloat16, device=dev)

    def pair():
        call('es_stem_conv_fwd', P(x), P(w), P(scale), P(shift), n_img, H, W, C, P(y), st)
        call('es_maxpool_fwd_h', P(y), C, P(nbr), n_img * Hp * Wp, 9, C, P(want), st)

    def fused():
        call('es_stem_pool_fwd', P(x), P(w), P(scale), P(shift), n_img, H, W, C, P(got), st)
    ts = []
    for fn in (pair, fused):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            fn()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 5 * 1e3)
    print(f'{case}: stem + pool as two launches {ts[0]:.1f} us, as one {ts[1]:.1f} us')
    assert torch.equal(got.view(torch.int16), want.view(torch.int16))
