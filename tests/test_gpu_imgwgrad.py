"""3x3 image weight gradient (csrc/imgwgrad.hip) on the GPU through the C ABI: against the map kernel (the oracle-pinned path) on the
same operands at the backbone's real shapes (tol 3e-6: only the f32 summation order differs), against f64 on a small case, run-to-run
bit-identical, and the engine's dispatch (the image backbone's conv2 weight gradients take it; ES_IMG_WGRAD off = map kernels, same
gradients to summation-order noise)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('case', [(8, 120, 120, 32, 1), (8, 60, 60, 64, 1), (3, 17, 45, 32, 1), (2, 9, 21, 64, 1), (8, 120, 120, 32, 2), (8, 60, 60, 64, 2),
                                  (2, 10, 22, 32, 2)])
def test_image_wgrad_vs_map_kernel(case):
    from embodiedscan_amd.engine import _wgrad as WG
    from embodiedscan_amd.hip import P, call, raw
    n_img, H, W, C, S = case
    Ho, Wo = H // S, W // S
    dev = torch.device('cuda:0')
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(5)
    n, n_o = n_img * H * W, n_img * Ho * Wo
    x = torch.randn(n, C, generator=g).to(dev)
    xh = x.to(torch.bfloat16)
    gy = torch.randn(n_o, C, generator=g).to(dev)
    nbr = torch.empty((n_o, 9), dtype=torch.int32, device=dev)
    call('es_image_map', n_img, H, W, Ho, Wo, 3, 3, S, 1, P(nbr), st)
    d1 = torch.zeros(9, C, C, device=dev)
    WG('es_spconv_wgrad_bf16_src', st, P(d1), P(xh), 1, C, P(gy), 0, C, P(nbr), n_o, n, 9, C, C)
    nf = int(raw('es_img_wgrad9_workspace_floats')(n_img, H, W, C, S))
    assert nf > 0
    ws = torch.full((nf,), float('nan'), device=dev)
    d2 = torch.full((9, C, C), float('nan'), device=dev)
    call('es_img_wgrad9_bf16', P(xh), C, P(gy), C, n_img, H, W, C, S, P(d2), 0, P(ws), nf, st)
    d3 = torch.full((9, C, C), float('nan'), device=dev)
    call('es_img_wgrad9_bf16', P(xh), C, P(gy), C, n_img, H, W, C, S, P(d3), 0, P(ws), nf, st)
    torch.cuda.synchronize()
    assert torch.equal(d2, d3), 'two runs differ'
    err = float((d1 - d2).abs().max() / d1.abs().max())
    print(f'{case}: image kernel vs map kernel max rel diff {err:.2e} (tol 3e-6)')
    assert err < 3e-6
    if n <= 4000:                                             # f64 on the bf16-rounded operands
        xr = xh.double().reshape(n_img, H, W, C)
        gr = gy.to(torch.bfloat16).double().reshape(n_img, Ho, Wo, C)
        xp = torch.zeros(n_img, H + 2, W + 2, C, dtype=torch.float64, device=dev)
        xp[:, 1:H + 1, 1:W + 1] = xr
        want = torch.stack([torch.einsum('nhwi,nhwo->io', xp[:, ty:ty + S * Ho:S, tx:tx + S * Wo:S], gr) for ty in range(3) for tx in range(3)])
        e64 = float((d2.double() - want).abs().max() / want.abs().max())
        print(f'   vs f64: {e64:.2e} (tol 2e-6)')
        assert e64 < 2e-6
    d4 = d1.clone()                                           # accumulate
    call('es_img_wgrad9_bf16', P(xh), C, P(gy), C, n_img, H, W, C, S, P(d4), 1, P(ws), nf, st)
    torch.cuda.synchronize()
    assert float((d4 - (d1 + d2)).abs().max() / d1.abs().max()) < 1e-6


def test_engine_takes_the_image_kernel_for_the_backbone_conv2():
    """one mv-3ddet train step: es_img_wgrad9_bf16 is called for the 32- / 64-channel conv2 layers, and with ES_IMG_WGRAD off the
    arena gradient agrees to summation-order noise"""
    import os
    from embodiedscan_amd import engine as E, hip, pipeline
    from embodiedscan_amd.config import build_detector, load_config
    from embodiedscan_amd.synth import make_scan
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    dev = torch.device('cuda:0')
    cfg = load_config(os.path.join(ROOT, 'configs', 'mv_3ddet.py'))
    scans = [make_scan(700 + i, n_views=3, height=240, width=320, img_size=(256, 256), n_points=20000) for i in range(2)]
    seen = []
    orig = hip._fn['es_img_wgrad9_bf16']

    def spy(*a):
        seen.append((a[4], a[5], a[6], a[7], a[8]))
        return orig(*a)
    saved = (E.PRECISION[0], E.IMG_WGRAD[0])
    grads = {}
    try:
        E.PRECISION[0] = 'bf16'
        hip._fn['es_img_wgrad9_bf16'] = spy
        for on in (True, False):
            E.IMG_WGRAD[0] = on
            det = build_detector(cfg, device=dev, seed=0).to(dev)
            batch = pipeline.make_batch([pipeline.upload_scan(s, dev) for s in scans])
            E.TAPE.clear()
            data = det.data_preprocessor(batch, True)
            det._bind()
            det.arena.grad.zero_()
            E.new_grad_epoch()
            det._tape_parts = []
            det.forward(data['inputs'], data['data_samples'], mode='loss')
            det._backward(None)
            torch.cuda.synchronize()
            grads[on] = (det.arena.grad[:det.arena.n_train].clone(), len(seen))
            seen_on = list(seen) if on else seen_on
            seen.clear()
            E.release(id(det))
    finally:
        hip._fn['es_img_wgrad9_bf16'] = orig
        E.PRECISION[0], E.IMG_WGRAD[0] = saved
        E.TAPE.clear()
    assert grads[True][1] > 0 and grads[False][1] == 0, (grads[True][1], grads[False][1])
    assert {c for *_, c, _s in seen_on} <= {32, 64} and {s_ for *_, s_ in seen_on} == {1, 2}
    a, b = grads[True][0], grads[False][0]
    err = float((a - b).norm() / b.norm())
    print(f'image kernel taken {grads[True][1]} times {sorted(set(seen_on))}; arena gradient vs map kernels: rel-L2 {err:.2e}')
    assert err < 1e-5


@pytest.mark.parametrize('case', [(288000, 32, 128), (288000, 128, 32), (72000, 64, 256), (72000, 256, 64), (18000, 128, 512), (18000, 512, 128),
                                  (1152000, 64, 32), (5000, 64, 64), (352224, 128, 320), (7528, 128, 320)])
def test_rows_wgrad_of_1x1_layers_vs_map_kernel(case):
    """es_rows_wgrad1_bf16 against the ring / gather kernel on the same operands (identity map), run-to-run bit-identical"""
    from embodiedscan_amd.engine import _wgrad as WG
    from embodiedscan_amd.hip import P, call, raw
    n, cin, cout = case
    dev = torch.device('cuda:0')
    st = torch.cuda.current_stream().cuda_stream
    raw('es_img_wgrad_set_option')(43, 0)                  # every width (the shipped rule: < 256 input channels only from 500 000 rows)
    g = torch.Generator().manual_seed(6)
    xh = torch.randn(n, cin, generator=g).to(dev).to(torch.bfloat16)
    gy = torch.randn(n, cout, generator=g).to(dev)
    d1 = torch.zeros(1, cin, cout, device=dev)
    WG('es_spconv_wgrad_bf16_src', st, P(d1), P(xh), 1, cin, P(gy), 0, cout, 0, n, n, 1, cin, cout)
    nf = int(raw('es_rows_wgrad1_workspace_floats')(n, cin, cout))
    assert nf > 0
    ws = torch.full((nf,), float('nan'), device=dev)
    d2 = torch.full((cin, cout), float('nan'), device=dev)
    call('es_rows_wgrad1_bf16', P(xh), cin, P(gy), cout, n, cin, cout, P(d2), 0, P(ws), nf, st)
    d3 = torch.full((cin, cout), float('nan'), device=dev)
    call('es_rows_wgrad1_bf16', P(xh), cin, P(gy), cout, n, cin, cout, P(d3), 0, P(ws), nf, st)
    torch.cuda.synchronize()
    assert torch.equal(d2, d3)
    err = float((d1[0] - d2).abs().max() / d1.abs().max())
    raw('es_img_wgrad_set_option')(43, 500000)
    print(f'{case}: rows kernel vs map kernel {err:.2e} (tol 5e-6), {nf // (cin * cout)} slices')
    assert err < 5e-6
