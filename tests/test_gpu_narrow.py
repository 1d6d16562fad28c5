"""MinkResNet.conv1's shape (K = 27 stride-2 map, 3 -> 64 channels, exact f32) on the lane-per-output-channel kernels of
csrc/spconv.hip (round 6) through the C ABI: forward and weight gradient against the tiled f32 kernels they replace (option 21 = 0:
the oracle-pinned path of tests/test_gpu_ops.py) on a real voxel set, run-to-run bit-identical; timings printed."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _time(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


@pytest.mark.parametrize('n_scans', [1, 4])
def test_conv1_kernels_vs_tiled_kernels(n_scans):
    from embodiedscan_amd import sparse, pipeline
    from embodiedscan_amd.engine import _wgrad as WG
    from embodiedscan_amd.hip import P, call, raw
    from embodiedscan_amd.synth import make_scan
    dev = torch.device('cuda:0')
    st = torch.cuda.current_stream().cuda_stream
    scans = [make_scan(77 + i, render_device='cuda:0') for i in range(n_scans)]
    pts = [pipeline.depth_to_points(pipeline.upload_scan(s, dev)) for s in scans]
    cs, _ = sparse.voxelize(pts, 0.01)
    o1 = cs.strided(2)
    nbr = cs.kernel_map(o1, 3)
    n_in, n_out, K, cin, cout = cs.n, o1.n, 27, 3, 64
    g = torch.Generator().manual_seed(5)
    x = torch.rand(n_in, cin, generator=g).to(dev)
    w = (torch.randn(K, cin, cout, generator=g) / 9).to(dev)
    bias = torch.randn(cout, generator=g).to(dev)
    dy = torch.randn(n_out, cout, generator=g).to(dev)
    out = {}
    for on in (1, 0):
        raw('es_set_option')(21, on)
        try:
            ys, dws = [], []
            for _ in range(2):
                y = torch.full((n_out, cout), float('nan'), device=dev)
                call('es_spconv_fwd', P(x), cin, P(w), P(nbr), n_out, n_in, K, cin, cout, P(bias), P(y), cout, 0, 0, st)
                dw = torch.zeros(K, cin, cout, device=dev)
                WG('es_spconv_wgrad', st, P(dw), P(x), cin, P(dy), cout, P(nbr), n_out, n_in, K, cin, cout)
                ys.append(y); dws.append(dw)
            torch.cuda.synchronize()
            assert torch.equal(ys[0], ys[1]) and torch.equal(dws[0], dws[1]), 'two runs differ'
            tf = _time(lambda: call('es_spconv_fwd', P(x), cin, P(w), P(nbr), n_out, n_in, K, cin, cout, P(bias), P(ys[1]), cout, 0, 0, st))
            tw = _time(lambda: WG('es_spconv_wgrad', st, P(dws[1]), P(x), cin, P(dy), cout, P(nbr), n_out, n_in, K, cin, cout))
            out[on] = (ys[0], dws[0], tf, tw)
        finally:
            raw('es_set_option')(21, 1)
    ef = float((out[1][0] - out[0][0]).abs().max() / out[0][0].abs().max())
    ew = float((out[1][1] - out[0][1]).abs().max() / out[0][1].abs().max())
    pairs = float((nbr >= 0).sum()) / n_out
    print(f'{n_scans} scans: {n_in} -> {n_out} rows, {pairs:.1f} pairs/row: forward {out[0][2]:.1f} -> {out[1][2]:.1f} us (rel diff {ef:.1e}), '
          f'weight gradient {out[0][3]:.1f} -> {out[1][3]:.1f} us (rel diff {ew:.1e})')
    assert ef < 1e-5 and ew < 1e-4
