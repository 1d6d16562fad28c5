"""Halo-tile K = 27 convolution (csrc/halo.hip) on the GPU, through the C ABI: the plan against numpy on the coordinate sets of a
synthetic scan, the convolution against the gather kernel (the oracle-pinned path of tests/test_gpu_config2.py) on the same
bf16 operands (tol 2e-5: only the f32 summation order differs) and against f64 on a sample of rows, run-to-run bit-identical,
the paged path on a scattered map, and the engine's dispatch (forward and data gradient of engine.conv take it for big maps)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
K = 27


@pytest.fixture(scope='module')
def sets():
    from embodiedscan_amd import pipeline, sparse
    from embodiedscan_amd.synth import make_scan
    dev = torch.device('cuda:0')
    scans = [make_scan(4321 + i, n_views=8, render_device='cuda:0') for i in range(2)]
    pts = [pipeline.depth_to_points(pipeline.upload_scan(s, dev)) for s in scans]
    cs, _ = sparse.voxelize(pts, 0.01)
    s8 = cs.strided(2).strided(2).strided(2)
    s32 = s8.strided(2).strided(2)
    L1 = s32.children()
    L0 = L1.children()
    return dict(s8=s8, L1=L1, L0=L0)


def _plan_np(nbr):
    n_out = nbr.shape[0]
    out = []
    for t0 in range(0, n_out, 256):
        blk = nbr[t0:t0 + 256]
        out.append(np.unique(blk[blk >= 0]))
    return out


def test_plan_matches_numpy_on_real_sets(sets):
    from embodiedscan_amd import engine as E
    for name in ('s8', 'L0'):
        S = sets[name]
        nbr = S.kernel_map(S, 3)
        loc, hrows, hcnt = E.halo_plan(nbr)
        torch.cuda.synchronize()
        nb, lc, hr, hc = nbr.cpu().numpy(), loc.cpu().numpy().view(np.uint16), hrows.cpu().numpy(), hcnt.cpu().numpy()
        want = _plan_np(nb)
        assert len(want) == len(hc)
        for t, u in enumerate(want):
            assert hc[t] == len(u) and np.array_equal(hr[t, :len(u)], u)
            blk = nb[t * 256:(t + 1) * 256]
            lt = lc[t * 256:t * 256 + len(blk)]
            assert np.array_equal(lt == 0xFFFF, blk < 0)
            assert np.array_equal(u[lt[blk >= 0]], blk[blk >= 0])
        assert (lc[nb.shape[0]:] == 0xFFFF).all()
        print(f'{name}: {S.n} rows, halo per 256-row tile mean {hc.mean():.0f} max {hc.max()} (resident 704)')
        assert hc.max() <= 704, 'Z-ordered sets are expected to fit the resident halo (the paged path would still be correct)'


@pytest.mark.parametrize('case', [('L0', 128, 128), ('L1', 256, 128), ('s8', 64, 256)])
def test_halo_convolution_vs_gather_kernel_and_f64(sets, case):
    from embodiedscan_amd import engine as E
    from embodiedscan_amd.hip import P, call
    name, cin, cout = case
    S = sets[name]
    dev = S.device
    n = S.n
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(3)
    nbr = S.kernel_map(S, 3)
    x = torch.randn(n, cin, generator=g).to(dev)
    xh = x.to(torch.bfloat16)
    w = (torch.randn(K, cin, cout, generator=g) / (K * cin) ** 0.5).to(dev)
    bias = torch.randn(cout, generator=g).to(dev)
    wn, wt = torch.empty((K, cin, cout), dtype=torch.bfloat16, device=dev), torch.empty((K, cout, cin), dtype=torch.bfloat16, device=dev)
    call('es_cast_weight_bf16', P(w), K, cin, cout, P(wn), P(wt), st)
    y1 = torch.empty(n, cout, device=dev)
    call('es_spconv_fwd_bf16', P(xh), 1, cin, P(wt), P(nbr), n, n, K, cin, cout, P(bias), P(y1), cout, 0, st)
    loc, hrows, hcnt = E.halo_plan(nbr)
    y2 = torch.full((n, cout), float('nan'), device=dev)
    call('es_spconv_halo_bf16', P(xh), cin, P(wt), P(loc), P(hrows), P(hcnt), n, n, K, cin, cout, P(bias), P(y2), cout, 0, 0, st)
    y3 = torch.full((n, cout), float('nan'), device=dev)
    call('es_spconv_halo_bf16', P(xh), cin, P(wt), P(loc), P(hrows), P(hcnt), n, n, K, cin, cout, P(bias), P(y3), cout, 0, 0, st)
    torch.cuda.synchronize()
    assert torch.equal(y2, y3), 'two runs of the halo kernel differ'
    err = float((y1 - y2).abs().max() / y1.abs().max())
    print(f'{name} {cin}->{cout} ({n} rows): halo vs gather kernel max rel diff {err:.2e} (tol 2e-5)')
    assert err < 2e-5
    # f64 on the bf16-rounded operands, 512 sampled rows
    rows = torch.randperm(n, generator=g)[:512].to(dev)
    nb = nbr[rows].long()
    xf, wf = xh.double(), wn.double()
    want = bias.double().expand(len(rows), cout).clone()
    for k in range(K):
        m = nb[:, k] >= 0
        want[m] += xf[nb[m, k]] @ wf[k]
    e64 = float((y2[rows].double() - want).abs().max() / want.abs().max())
    print(f'   vs f64 on the rounded operands: {e64:.2e} (tol 2e-6)')
    assert e64 < 2e-6
    # accumulate into a strided output
    y4 = torch.randn(n, cout + 8, device=dev)
    y40 = y4.clone()
    call('es_spconv_halo_bf16', P(xh), cin, P(wt), P(loc), P(hrows), P(hcnt), n, n, K, cin, cout, 0, P(y4), cout + 8, 1, 0, st)
    torch.cuda.synchronize()
    assert torch.equal(y4[:, cout:], y40[:, cout:])
    ea = float((y4[:, :cout] - (y40[:, :cout] + y2 - bias)).abs().max() / y1.abs().max())
    assert ea < 1e-5, ea


def test_paged_halo_on_a_scattered_map():
    from embodiedscan_amd import engine as E
    from embodiedscan_amd.hip import P, call
    dev = torch.device('cuda:0')
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(11)
    n_out, n_in, cin, cout = 3000, 50000, 128, 128
    nbr = torch.randint(0, n_in, (n_out, K), generator=g, dtype=torch.int32)
    nbr[torch.rand(n_out, K, generator=g) < 0.5] = -1
    nbr = nbr.to(dev)
    x = torch.randn(n_in, cin, generator=g).to(dev)
    xh = x.to(torch.bfloat16)
    w = (torch.randn(K, cin, cout, generator=g) / (K * cin) ** 0.5).to(dev)
    wn, wt = torch.empty((K, cin, cout), dtype=torch.bfloat16, device=dev), torch.empty((K, cout, cin), dtype=torch.bfloat16, device=dev)
    call('es_cast_weight_bf16', P(w), K, cin, cout, P(wn), P(wt), st)
    loc, hrows, hcnt = E.halo_plan(nbr)
    assert int(hcnt.max()) > 2 * 704
    y1, y2 = torch.empty(n_out, cout, device=dev), torch.empty(n_out, cout, device=dev)
    call('es_spconv_fwd_bf16', P(xh), 1, cin, P(wt), P(nbr), n_out, n_in, K, cin, cout, 0, P(y1), cout, 0, st)
    call('es_spconv_halo_bf16', P(xh), cin, P(wt), P(loc), P(hrows), P(hcnt), n_out, n_in, K, cin, cout, 0, P(y2), cout, 0, 0, st)
    torch.cuda.synchronize()
    err = float((y1 - y2).abs().max() / y1.abs().max())
    print(f'paged halo ({int(hcnt.max())} distinct source rows in a tile): max rel diff vs gather kernel {err:.2e}')
    assert err < 2e-5


def test_engine_conv_takes_the_halo_kernel_for_big_maps(sets):
    """engine.conv forward + backward on the L0 set: the halo entry point is called for the forward and for the data gradient
    (inverse map), and the results agree with ES_HALO off (gather kernels) to summation-order noise."""
    from embodiedscan_amd import engine as E, hip
    S = sets['L0']
    dev = S.device
    n, cin, cout = S.n, 128, 128
    nbr, inv = S.kernel_map(S, 3), S.inverse_map(S, 3)
    g = torch.Generator().manual_seed(5)
    xd = torch.randn(n, cin, generator=g).to(dev)
    wd = (torch.randn(K, cin, cout, generator=g) / (K * cin) ** 0.5).to(dev)
    gy = torch.randn(n, cout, generator=g).to(dev)
    saved = (E.PRECISION[0], E.HALO[0])
    res = {}
    seen = []
    orig = hip._fn['es_spconv_halo_bf16']

    def spy(*a):
        seen.append(1)
        return orig(*a)
    try:
        E.PRECISION[0] = 'bf16'
        hip.raw('es_halo_set_option')(30, 1)            # (the test's sets are small: take the halo kernel whatever the workgroup count)
        for halo in (True, False):
            E.HALO[0] = halo
            hip._fn['es_spconv_halo_bf16'] = spy
            E.TAPE.clear()
            x = E.Var(xd.clone())
            w = E.Param(wd.clone(), torch.zeros_like(wd))
            y = E.conv(x, w, nbr, inv, n)
            y.g = gy.clone()
            E.TAPE.backward()
            E.join_wgrad_streams()
            torch.cuda.synchronize()
            res[halo] = (y.d.clone(), x.g.clone(), w.g.clone(), len(seen))
            seen.clear()
    finally:
        hip._fn['es_spconv_halo_bf16'] = orig
        hip.raw('es_halo_set_option')(30, 192)
        E.PRECISION[0], E.HALO[0] = saved
        E.TAPE.clear()
    assert res[True][3] == 2 and res[False][3] == 0, (res[True][3], res[False][3])
    # the data gradient ran on the FORWARD map's plan (mirrored taps): the inverse map carries no plan of its own, and it IS the flipped map
    assert getattr(inv, '_mirror_of', None) is nbr and getattr(inv, '_halo', None) is None and getattr(nbr, '_halo', None) is not None
    assert torch.equal(inv, nbr.flip(1))
    for i, what in enumerate(('output', 'data gradient', 'weight gradient')):
        a, b = res[True][i], res[False][i]
        err = float((a - b).abs().max() / b.abs().max())
        print(f'engine.conv halo vs gather, {what}: max rel diff {err:.2e}')
        assert err < 2e-5
