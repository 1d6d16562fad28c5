"""Few-row linear layers (the grounding decoder's 256 -> 256 / 256 <-> 2 048 GEMMs on 3 072 query rows, 396 text rows) on the whole-stage
64 x 64 kernels of csrc/spconv.hip (round 6, option 24) through the C ABI: forward / data gradient bit for bit against the 128-row kernel
they replace, weight gradient against the gather-kernel family within 5e-6, run-to-run bit-identical; timings printed."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _time(fn, n=20):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


@pytest.mark.parametrize('case', [(3072, 256, 256), (3072, 2048, 256), (3072, 256, 2048), (396, 256, 256), (3072, 256, 64), (1001, 520, 128)])
def test_forward_vs_row_gemm(case):
    from embodiedscan_amd.hip import P, call, raw
    n, cin, cout = case
    dev = torch.device('cuda:0')
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(2)
    x = torch.randn(n, cin, generator=g).to(dev)
    w = (torch.randn(1, cin, cout, generator=g) / cin ** 0.5).to(dev)
    bias = torch.randn(cout, generator=g).to(dev)
    wn, wt = torch.empty((1, cin, cout), dtype=torch.bfloat16, device=dev), torch.empty((1, cout, cin), dtype=torch.bfloat16, device=dev)
    call('es_cast_weight_bf16', P(w), 1, cin, cout, P(wn), P(wt), st)
    ys, ts = {}, {}
    try:
        for on in (256, 0):
            raw('es_set_option')(24, on)
            y = torch.full((n, cout), float('nan'), device=dev)
            call('es_spconv_fwd_bf16', P(x), 0, cin, P(wt), 0, n, n, 1, cin, cout, P(bias), P(y), cout, 0, st)
            y2 = torch.full((n, cout), float('nan'), device=dev)
            call('es_spconv_fwd_bf16', P(x), 0, cin, P(wt), 0, n, n, 1, cin, cout, P(bias), P(y2), cout, 0, st)
            torch.cuda.synchronize()
            assert torch.equal(y, y2)
            ts[on] = _time(lambda: call('es_spconv_fwd_bf16', P(x), 0, cin, P(wt), 0, n, n, 1, cin, cout, P(bias), P(y2), cout, 0, st))
            ys[on] = y
    finally:
        raw('es_set_option')(24, 256)
    ref = x.to(torch.bfloat16).double() @ w[0].to(torch.bfloat16).double() + bias.double()
    err = float((ys[256].double() - ref).abs().max() / ref.abs().max())
    same = bool(torch.equal(ys[256], ys[0]))
    print(f'{case}: 128-row kernel {ts[0]:.1f} us, whole-stage kernel {ts[256]:.1f} us; vs f64 on the rounded operands {err:.1e}; same bits as the 128-row kernel: {same}')
    assert err < 2e-6 and float((ys[256] - ys[0]).abs().max() / ref.abs().max()) < 1e-6


@pytest.mark.parametrize('case', [(3072, 256, 256), (396, 256, 256), (3072, 128, 64), (777, 64, 256)])
def test_weight_gradient_vs_gather_family(case):
    from embodiedscan_amd.engine import _wgrad as WG
    from embodiedscan_amd.hip import P, raw
    n, cin, cout = case
    dev = torch.device('cuda:0')
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(3)
    x = torch.randn(n, cin, generator=g).to(dev)
    gy = torch.randn(n, cout, generator=g).to(dev)
    ds, ts = {}, {}
    try:
        for on in (256, 0):
            raw('es_set_option')(24, on)
            d1, d2 = torch.zeros(1, cin, cout, device=dev), torch.zeros(1, cin, cout, device=dev)
            WG('es_spconv_wgrad_bf16', st, P(d1), P(x), cin, P(gy), cout, 0, n, n, 1, cin, cout)
            WG('es_spconv_wgrad_bf16', st, P(d2), P(x), cin, P(gy), cout, 0, n, n, 1, cin, cout)
            torch.cuda.synchronize()
            assert torch.equal(d1, d2), 'two runs differ'
            d1b = d1.clone()
            ts[on] = _time(lambda: WG('es_spconv_wgrad_bf16', st, P(d2), P(x), cin, P(gy), cout, 0, n, n, 1, cin, cout))
            ds[on] = d1b
    finally:
        raw('es_set_option')(24, 256)
    ref = x.to(torch.bfloat16).double().T @ gy.to(torch.bfloat16).double()
    err = float((ds[256][0].double() - ref).abs().max() / ref.abs().max())
    print(f'{case}: gather-family kernel {ts[0]:.1f} us, 64 x 64 tile kernel {ts[256]:.1f} us (both incl. the slice reduction); vs f64 {err:.1e}')
    assert err < 5e-6 and float((ds[256] - ds[0]).abs().max() / ref.abs().max()) < 5e-6


@pytest.mark.parametrize('case', [(1152000, 16, 1), (288000, 32, 1), (72000, 64, 1), (100003, 32, 0), (3456000, 16, 1)])
def test_expansion_layers_stream_kernel_vs_row_gemm(case):
    """the image backbone's 1x1 C -> 4 C layers (bf16 rows, frozen BN, bf16 residual, ReLU): k_expand_bf16 against k_rowgemm2_bf16"""
    from embodiedscan_amd.hip import P, call, raw
    n, cin, with_res = case
    cout = 4 * cin
    dev = torch.device('cuda:0')
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(7)
    x = torch.randn(n, cin, generator=g).to(dev).to(torch.bfloat16)
    res = torch.randn(n, cout, generator=g).to(dev).to(torch.bfloat16)
    w = (torch.randn(1, cin, cout, generator=g) / cin ** 0.5).to(dev)
    scale, shift = (0.5 + torch.rand(cout, generator=g)).to(dev), torch.randn(cout, generator=g).to(dev)
    wn, wt = torch.empty((1, cin, cout), dtype=torch.bfloat16, device=dev), torch.empty((1, cout, cin), dtype=torch.bfloat16, device=dev)
    call('es_cast_weight_bf16', P(w), 1, cin, cout, P(wn), P(wt), st)
    ys, ts = {}, {}

    def run(y):
        call('es_spconv_fwd_bf16_io', P(x), 1, cin, P(wt), 0, n, n, 1, cin, cout, P(scale), P(shift), P(res) if with_res else 0, 1, cout, 1, P(y), 1, cout, st)
    try:
        for on in (65536, 0):
            raw('es_set_option')(25, on)
            y, y2 = torch.full((n, cout), float('nan'), dtype=torch.bfloat16, device=dev), torch.full((n, cout), float('nan'), dtype=torch.bfloat16, device=dev)
            run(y); run(y2)
            torch.cuda.synchronize()
            assert torch.equal(y.view(torch.int16), y2.view(torch.int16))
            ts[on] = _time(lambda: run(y2))
            ys[on] = y
    finally:
        raw('es_set_option')(25, 65536)
    mb = n * (cin * 2 + cout * 2 * (2 if with_res else 1)) / 1e6
    ulp = (ys[65536].view(torch.int16).int() - ys[0].view(torch.int16).int()).abs()
    print(f'{case}: {mb:.0f} MB | row GEMM {ts[0]:.1f} us ({mb / ts[0]:.2f} TB/s) | stream kernel {ts[65536]:.1f} us ({mb / ts[65536]:.2f} TB/s) | differing bf16 values {int((ulp > 0).sum())}')
    assert int(ulp.max()) == 0
