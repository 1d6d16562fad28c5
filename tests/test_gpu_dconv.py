"""The dense-volume convolution engine (csrc/dconv.hip; nn.Conv3d of IndoorImVoxelNeck, embodiedscan/models/necks/imvoxel_neck.py:
78-143, by address arithmetic) on the MI355X:
  * small volumes against an f64 evaluation of nn.Conv3d's arithmetic on the bf16-rounded operands (every border case, stride 2,
    batch > 1, both row tiles, both loop orders, forced slice counts, accumulation) -- forward, stride-1 data gradient, weight
    gradient (2e-6 of the output scale: f32 accumulation of exact bf16 products);
  * the occupancy neck's real shapes (40x40x16 x 768, 20x20x8 x 1536, 10x10x4 x 3072) against the neighbour-map kernels the
    oracle-pinned tests hold (same operands, same precision, different order of the f32 additions: 2e-5);
  * run-to-run bit-reproducibility of split launches."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref_conv(xb, wb, B, X, Y, Z, ks, st, pad):
    """f64 nn.Conv3d on channels-last rows, taps ordered (kx, ky, kz): xb (B*X*Y*Z, Cin) f64, wb (K, Cin, Cout) f64"""
    o = lambda d: (d + 2 * pad - ks) // st + 1
    Xo, Yo, Zo = o(X), o(Y), o(Z)
    cin, cout = wb.shape[1], wb.shape[2]
    xv = torch.zeros((B, X + 2 * pad, Y + 2 * pad, Z + 2 * pad, cin), dtype=torch.float64, device=xb.device)
    xv[:, pad:pad + X, pad:pad + Y, pad:pad + Z] = xb.reshape(B, X, Y, Z, cin)
    y = torch.zeros((B, Xo, Yo, Zo, cout), dtype=torch.float64, device=xb.device)
    for kx in range(ks):
        for ky in range(ks):
            for kz in range(ks):
                y += xv[:, kx:kx + st * Xo:st, ky:ky + st * Yo:st, kz:kz + st * Zo:st] @ wb[(kx * ks + ky) * ks + kz]
    return y.reshape(-1, cout), (Xo, Yo, Zo)


def _geom(*v):
    return (ctypes.c_int * 7)(*v)


def _ws(hip, g, mode, cin, cout, dev):
    nf = int(hip.raw('es_dconv_workspace_floats')(g, mode, cin, cout))
    return torch.empty(max(nf, 4), device=dev), nf


def test_dense_engine_small_volumes_vs_f64():
    from embodiedscan_amd import hip
    from embodiedscan_amd.hip import call, P
    dev = torch.device('cuda:0')
    st_ = torch.cuda.current_stream().cuda_stream
    gen = torch.Generator().manual_seed(3)
    opt = hip.raw('es_dconv_set_option')
    try:
        for B, X, Y, Z, st, cin, cout in ((1, 7, 6, 5, 1, 64, 256), (2, 9, 8, 5, 1, 256, 256), (1, 10, 8, 6, 2, 128, 512),
                                          (1, 12, 11, 9, 1, 256, 512), (2, 11, 9, 6, 1, 192, 128), (1, 40, 40, 16, 1, 768, 128)):
            g = _geom(B, X, Y, Z, 3, st, 1)
            x = torch.randn(B * X * Y * Z, cin, generator=gen).to(dev)
            w = (torch.randn(27, cin, cout, generator=gen) / (27 * cin) ** 0.5).to(dev)
            xh = x.bfloat16().contiguous()
            wt = torch.empty((27, cout, cin), dtype=torch.bfloat16, device=dev)
            wn = torch.empty((27, cin, cout), dtype=torch.bfloat16, device=dev)
            call('es_cast_weight_bf16', P(w), 27, cin, cout, P(wn), P(wt), st_)
            xb, wb = xh.double(), wn.double()
            want, (Xo, Yo, Zo) = _ref_conv(xb, wb, B, X, Y, Z, 3, st, 1)
            M = B * Xo * Yo * Zo
            scale = want.abs().max()
            for rows, order, split in ((0, 1, 0), (256, 0, 0), (320, 1, 3), (256, 1, 5), (320, 0, 2)):
                opt(20, rows); opt(21, order); opt(22, split)
                ws, nf = _ws(hip, g, 0, cin, cout, dev)
                y = torch.full((M, cout), float('nan'), device=dev)
                call('es_dconv_fwd_bf16', P(xh), cin, P(wt), g, 0, cin, cout, P(y), cout, 0, P(ws), nf, st_)
                err = float((y.double() - want).abs().max() / scale)
                assert err < 2e-6, ('fwd', B, X, Y, Z, st, rows, order, split, err)
                y2 = torch.full((M, cout), float('nan'), device=dev)
                call('es_dconv_fwd_bf16', P(xh), cin, P(wt), g, 0, cin, cout, P(y2), cout, 0, P(ws), nf, st_)
                assert torch.equal(y, y2)                            # slice-ordered reduction: bit-reproducible
            y3 = torch.ones((M, cout), device=dev)
            call('es_dconv_fwd_bf16', P(xh), cin, P(wt), g, 0, cin, cout, P(y3), cout, 1, P(ws), nf, st_)
            assert float((y3.double() - 1 - want).abs().max() / scale) < 2e-6
            opt(20, 0); opt(21, 0); opt(22, 0)
            dy = torch.randn(M, cout, generator=gen).to(dev)
            dyh = dy.bfloat16().contiguous()
            if cin % 256 == 0 and cout % 256 == 0:
                # weight gradient: dW[k] = X[src(., k)]^T dY
                xv = torch.zeros((B, X + 2, Y + 2, Z + 2, cin), dtype=torch.float64, device=dev)
                xv[:, 1:1 + X, 1:1 + Y, 1:1 + Z] = xb.reshape(B, X, Y, Z, cin)
                wantw = torch.stack([xv[:, kx:kx + st * Xo:st, ky:ky + st * Yo:st, kz:kz + st * Zo:st].reshape(M, cin).T @ dyh.double()
                                     for kx in range(3) for ky in range(3) for kz in range(3)])
                dw = torch.full((27, cin, cout), float('nan'), device=dev)
                call('es_dconv_wgrad_bf16', P(xh), cin, P(dyh), cout, g, 0, cin, cout, P(dw), 0, st_)
                assert float((dw.double() - wantw).abs().max() / wantw.abs().max()) < 2e-6, ('wgrad', B, X, Y, Z, st)
                dw2 = torch.ones((27, cin, cout), device=dev)
                call('es_dconv_wgrad_bf16', P(xh), cin, P(dyh), cout, g, 0, cin, cout, P(dw2), 1, st_)
                assert float((dw2.double() - 1 - wantw).abs().max() / wantw.abs().max()) < 2e-6
            if st == 1 and cin % 256 == 0:
                # data gradient = the adjoint: conv of dY with the mirrored taps and W^T
                wf = wn.double().flip(0).transpose(1, 2).contiguous()
                wantx, _ = _ref_conv(dyh.double(), wf, B, X, Y, Z, 3, 1, 1)
                ws, nf = _ws(hip, g, 1, cin, cout, dev)
                dx = torch.full((B * X * Y * Z, cin), float('nan'), device=dev)
                call('es_dconv_fwd_bf16', P(dyh), cout, P(wn), g, 1, cin, cout, P(dx), cin, 0, P(ws), nf, st_)
                assert float((dx.double() - wantx).abs().max() / wantx.abs().max()) < 2e-6, ('dgrad', B, X, Y, Z)
    finally:
        opt(20, 0); opt(21, 0); opt(22, 0)


@pytest.mark.parametrize('X,Y,Z,cin,cout,st', [(40, 40, 16, 768, 768, 1), (40, 40, 16, 768, 1536, 2), (20, 20, 8, 1536, 1536, 1),
                                                (10, 10, 4, 3072, 3072, 1)])
def test_dense_engine_neck_shapes_vs_map_kernels(X, Y, Z, cin, cout, st):
    """config-5 shapes: the dense engine against the neighbour-map launches on the same bf16 operands"""
    from embodiedscan_amd import hip
    from embodiedscan_amd.hip import call, P
    from embodiedscan_amd.models.necks.imvoxel_neck import VolumeGrid
    dev = torch.device('cuda:0')
    st_ = torch.cuda.current_stream().cuda_stream
    gen = torch.Generator().manual_seed(11)
    g = _geom(1, X, Y, Z, 3, st, 1)
    grid = VolumeGrid(1, X, Y, Z, dev)
    nbr, inv, n_out, _ = grid.conv_map(3, st, 1)
    n_in = X * Y * Z
    x = torch.randn(n_in, cin, generator=gen).to(dev)
    w = (torch.randn(27, cin, cout, generator=gen) / (27 * cin) ** 0.5).to(dev)
    xh = x.bfloat16().contiguous()
    wt = torch.empty((27, cout, cin), dtype=torch.bfloat16, device=dev)
    wn = torch.empty((27, cin, cout), dtype=torch.bfloat16, device=dev)
    call('es_cast_weight_bf16', P(w), 27, cin, cout, P(wn), P(wt), st_)
    # forward
    ws, nf = _ws(hip, g, 0, cin, cout, dev)
    y = torch.empty(n_out, cout, device=dev)
    call('es_dconv_fwd_bf16', P(xh), cin, P(wt), g, 0, cin, cout, P(y), cout, 0, P(ws), nf, st_)
    nfm = int(hip.raw('es_spconv_split_workspace_floats')(n_out, 27, cin, cout))
    wsm = torch.zeros(max(nfm, 4), device=dev)
    y0 = torch.empty(n_out, cout, device=dev)
    if nfm:
        call('es_spconv_fwd_bf16_ws', P(xh), 1, cin, P(wt), P(nbr), n_out, n_in, 27, cin, cout, 0, P(y0), cout, 0, P(wsm), nfm, st_)
    else:
        call('es_spconv_fwd_bf16', P(xh), 1, cin, P(wt), P(nbr), n_out, n_in, 27, cin, cout, 0, P(y0), cout, 0, st_)
    err = float((y - y0).abs().max() / y0.abs().max())
    assert err < 2e-5, ('fwd', err)
    # weight gradient
    dy = torch.randn(n_out, cout, generator=gen).to(dev)
    dyh = dy.bfloat16().contiguous()
    dw = torch.empty(27, cin, cout, device=dev)
    call('es_dconv_wgrad_bf16', P(xh), cin, P(dyh), cout, g, 0, cin, cout, P(dw), 0, st_)
    need = int(hip.raw('es_spconv_wgrad_workspace_floats')(1, P(xh), 1, cin, P(dyh), 1, cout, n_out, n_in, 27, cin, cout))
    wsw = torch.empty(max(need, 4), device=dev)
    dw0 = torch.empty(27, cin, cout, device=dev)
    call('es_spconv_wgrad_bf16_src', P(xh), 1, cin, P(dyh), 1, cout, P(nbr), n_out, n_in, 27, cin, cout, P(dw0), 0, P(wsw), need, st_)
    err = float((dw - dw0).abs().max() / dw0.abs().max())
    assert err < 2e-5, ('wgrad', err)
    if st == 1:
        ws, nf = _ws(hip, g, 1, cin, cout, dev)
        dx = torch.empty(n_in, cin, device=dev)
        call('es_dconv_fwd_bf16', P(dyh), cout, P(wn), g, 1, cin, cout, P(dx), cin, 0, P(ws), nf, st_)
        nfm = int(hip.raw('es_spconv_split_workspace_floats')(n_in, 27, cout, cin))
        wsm = torch.zeros(max(nfm, 4), device=dev)
        dx0 = torch.empty(n_in, cin, device=dev)
        if nfm:
            call('es_spconv_fwd_bf16_ws', P(dyh), 1, cout, P(wn), P(inv), n_in, n_out, 27, cout, cin, 0, P(dx0), cin, 0, P(wsm), nfm, st_)
        else:
            call('es_spconv_fwd_bf16', P(dyh), 1, cout, P(wn), P(inv), n_in, n_out, 27, cout, cin, 0, P(dx0), cin, 0, st_)
        err = float((dx - dx0).abs().max() / dx0.abs().max())
        assert err < 2e-5, ('dgrad', err)


def _classes(t, B, X, Y, Z, C):
    """(B*2X*2Y*2Z, C) rows -> (8, B*X*Y*Z, C): class p = (px, py, pz) holds the rows of the voxels (2x+px, 2y+py, 2z+pz)"""
    v = t.reshape(B, X, 2, Y, 2, Z, 2, C)
    return torch.stack([v[:, :, px, :, py, :, pz].reshape(-1, C) for px in range(2) for py in range(2) for pz in range(2)])


def test_parity_class_launches_small_volumes_vs_f64():
    """data gradient of nn.Conv3d(k=3, s=2, p=1) and nn.ConvTranspose3d(k=2, s=2) forward / data gradient / weight gradient on the
    dense engine against f64 adjoints on the bf16-rounded operands (2e-6)"""
    from embodiedscan_amd import hip
    from embodiedscan_amd.hip import call, P
    dev = torch.device('cuda:0')
    st_ = torch.cuda.current_stream().cuda_stream
    gen = torch.Generator().manual_seed(13)
    opt = hip.raw('es_dconv_set_option')
    try:
        for B, Xo, Yo, Zo, cin, cout in ((1, 5, 4, 3, 256, 128), (2, 7, 6, 5, 512, 256), (1, 20, 20, 8, 256, 64)):
            X, Y, Z = 2 * Xo, 2 * Yo, 2 * Zo
            g = _geom(B, X, Y, Z, 3, 2, 1)
            assert hip.raw('es_dconv_supported')(g, 1, cin, cout) == 1
            w = (torch.randn(27, cin, cout, generator=gen) / (27 * cout) ** 0.5).to(dev)
            wt = torch.empty((27, cout, cin), dtype=torch.bfloat16, device=dev)
            wn = torch.empty((27, cin, cout), dtype=torch.bfloat16, device=dev)
            call('es_cast_weight_bf16', P(w), 27, cin, cout, P(wn), P(wt), st_)
            M = B * Xo * Yo * Zo
            dyh = torch.randn(M, cout, generator=gen).to(dev).bfloat16().contiguous()
            dyb, wb = dyh.double(), wn.double()
            dxp = torch.zeros((B, X + 2, Y + 2, Z + 2, cin), dtype=torch.float64, device=dev)
            for kx in range(3):
                for ky in range(3):
                    for kz in range(3):
                        dxp[:, kx:kx + 2 * Xo:2, ky:ky + 2 * Yo:2, kz:kz + 2 * Zo:2] += (dyb @ wb[(kx * 3 + ky) * 3 + kz].T).reshape(B, Xo, Yo, Zo, cin)
            want = dxp[:, 1:1 + X, 1:1 + Y, 1:1 + Z].reshape(-1, cin)
            for rows, split in ((0, 0), (256, 1), (320, 1), (0, 2), (320, 5)):      # split > 1: slices of every class through the workspace
                opt(20, rows); opt(22, split)
                ws, nf = _ws(hip, g, 1, cin, cout, dev)
                dx = torch.full((B * X * Y * Z, cin), float('nan'), device=dev)
                call('es_dconv_fwd_bf16', P(dyh), cout, P(wn), g, 1, cin, cout, P(dx), cin, 0, P(ws), nf, st_)
                err = float((dx.double() - want).abs().max() / want.abs().max())
                assert err < 2e-6, ('strided dgrad', B, Xo, Yo, Zo, rows, split, err)
                dx2 = torch.ones((B * X * Y * Z, cin), device=dev)
                call('es_dconv_fwd_bf16', P(dyh), cout, P(wn), g, 1, cin, cout, P(dx2), cin, 1, P(ws), nf, st_)
                assert float((dx2.double() - 1 - want).abs().max() / want.abs().max()) < 2e-6
            opt(20, 0); opt(22, 0)
            # transposed convolution on the coarse grid (Xo, Yo, Zo)
            ci_t, co_t = 256, 256
            gt = _geom(B, Xo, Yo, Zo, 2, 2, 0)
            assert all(hip.raw('es_dconv_supported')(gt, m, ci_t, co_t) == 1 for m in (3, 4, 5))
            xh = torch.randn(M, ci_t, generator=gen).to(dev).bfloat16().contiguous()
            w8 = (torch.randn(8, ci_t, co_t, generator=gen) / ci_t ** 0.5).to(dev)
            w8t = torch.empty((8, co_t, ci_t), dtype=torch.bfloat16, device=dev)
            w8n = torch.empty((8, ci_t, co_t), dtype=torch.bfloat16, device=dev)
            call('es_cast_weight_bf16', P(w8), 8, ci_t, co_t, P(w8n), P(w8t), st_)
            y = torch.full((8 * M, co_t), float('nan'), device=dev)
            ws, nf = _ws(hip, gt, 3, ci_t, co_t, dev)
            call('es_dconv_fwd_bf16', P(xh), ci_t, P(w8t), gt, 3, ci_t, co_t, P(y), co_t, 0, P(ws), nf, st_)
            got = _classes(y.double(), B, Xo, Yo, Zo, co_t)
            wantf = torch.stack([xh.double() @ w8n[p].double() for p in range(8)])
            assert float((got - wantf).abs().max() / wantf.abs().max()) < 2e-6, 'transposed fwd'
            dyf = torch.randn(8 * M, co_t, generator=gen).to(dev).bfloat16().contiguous()
            cls = _classes(dyf.double(), B, Xo, Yo, Zo, co_t)
            wantx = sum(cls[p] @ w8n[p].double().T for p in range(8))
            ws, nf = _ws(hip, gt, 4, ci_t, co_t, dev)
            dxc = torch.full((M, ci_t), float('nan'), device=dev)
            call('es_dconv_fwd_bf16', P(dyf), co_t, P(w8n), gt, 4, ci_t, co_t, P(dxc), ci_t, 0, P(ws), nf, st_)
            assert float((dxc.double() - wantx).abs().max() / wantx.abs().max()) < 2e-6, 'transposed dgrad'
            wantw = torch.stack([xh.double().T @ cls[p] for p in range(8)])
            dw = torch.full((8, ci_t, co_t), float('nan'), device=dev)
            call('es_dconv_wgrad_bf16', P(xh), ci_t, P(dyf), co_t, gt, 1, ci_t, co_t, P(dw), 0, st_)
            assert float((dw.double() - wantw).abs().max() / wantw.abs().max()) < 2e-6, 'transposed wgrad'
    finally:
        opt(20, 0); opt(22, 0)


@pytest.mark.parametrize('X,Y,Z,cin,cout', [(40, 40, 16, 768, 1536), (20, 20, 8, 1536, 3072)])
def test_strided_data_gradient_neck_shapes_vs_map_kernel(X, Y, Z, cin, cout):
    from embodiedscan_amd import hip
    from embodiedscan_amd.hip import call, P
    from embodiedscan_amd.models.necks.imvoxel_neck import VolumeGrid
    dev = torch.device('cuda:0')
    st_ = torch.cuda.current_stream().cuda_stream
    gen = torch.Generator().manual_seed(17)
    g = _geom(1, X, Y, Z, 3, 2, 1)
    nbr, inv, n_out, _ = VolumeGrid(1, X, Y, Z, dev).conv_map(3, 2, 1)
    n_in = X * Y * Z
    w = (torch.randn(27, cin, cout, generator=gen) / (27 * cout) ** 0.5).to(dev)
    wt = torch.empty((27, cout, cin), dtype=torch.bfloat16, device=dev)
    wn = torch.empty((27, cin, cout), dtype=torch.bfloat16, device=dev)
    call('es_cast_weight_bf16', P(w), 27, cin, cout, P(wn), P(wt), st_)
    dyh = torch.randn(n_out, cout, generator=gen).to(dev).bfloat16().contiguous()
    dx = torch.empty(n_in, cin, device=dev)
    ws, nf = _ws(hip, g, 1, cin, cout, dev)
    call('es_dconv_fwd_bf16', P(dyh), cout, P(wn), g, 1, cin, cout, P(dx), cin, 0, P(ws), nf, st_)
    nfm = int(hip.raw('es_spconv_split_workspace_floats')(n_in, 27, cout, cin))
    wsm = torch.zeros(max(nfm, 4), device=dev)
    dx0 = torch.empty(n_in, cin, device=dev)
    if nfm:
        call('es_spconv_fwd_bf16_ws', P(dyh), 1, cout, P(wn), P(inv), n_in, n_out, 27, cout, cin, 0, P(dx0), cin, 0, P(wsm), nfm, st_)
    else:
        call('es_spconv_fwd_bf16', P(dyh), 1, cout, P(wn), P(inv), n_in, n_out, 27, cout, cin, 0, P(dx0), cin, 0, st_)
    err = float((dx - dx0).abs().max() / dx0.abs().max())
    assert err < 2e-5, err


@pytest.mark.gpu
@pytest.mark.parametrize('X,Y,Z,cin,cout', [(40, 40, 16, 768, 1536), (20, 20, 8, 1536, 3072)])
def test_pointwise_stride2_downsample_neck_shapes_vs_f32(X, Y, Z, cin, cout):
    """the neck's identity down-sample nn.Conv3d(k=1, s=2): forward, data gradient (parity classes, class 0 alone has a tap) and
    weight gradient of the dense engine against f32 matrix products on the bf16-rounded operands.  Tolerance 2e-5 of the largest
    magnitude (f32 accumulation order only)."""
    from embodiedscan_amd import hip
    from embodiedscan_amd.hip import call, P
    dev = torch.device('cuda:0')
    st_ = torch.cuda.current_stream().cuda_stream
    gen = torch.Generator().manual_seed(23)
    g = _geom(1, X, Y, Z, 1, 2, 0)
    for mode in (0, 1, 2):
        assert hip.raw('es_dconv_supported')(g, mode, cin, cout) == 1
    n_in, M = X * Y * Z, (X // 2) * (Y // 2) * (Z // 2)
    x = torch.randn(n_in, cin, generator=gen).to(dev)
    w = (torch.randn(1, cin, cout, generator=gen) / cin ** 0.5).to(dev)
    wt = torch.empty((1, cout, cin), dtype=torch.bfloat16, device=dev)
    wn = torch.empty((1, cin, cout), dtype=torch.bfloat16, device=dev)
    call('es_cast_weight_bf16', P(w), 1, cin, cout, P(wn), P(wt), st_)
    xh = x.bfloat16().contiguous()
    xs = xh.float().view(X, Y, Z, cin)[::2, ::2, ::2].reshape(M, cin)
    wb = wn.float()[0]
    y = torch.empty(M, cout, device=dev)
    ws, nf = _ws(hip, g, 0, cin, cout, dev)
    call('es_dconv_fwd_bf16', P(xh), cin, P(wt), g, 0, cin, cout, P(y), cout, 0, P(ws), nf, st_)
    want = xs @ wb
    assert float((y - want).abs().max() / want.abs().max()) < 2e-5
    dyh = torch.randn(M, cout, generator=gen).to(dev).bfloat16().contiguous()
    dx = torch.full((n_in, cin), float('nan'), device=dev)
    ws, nf = _ws(hip, g, 1, cin, cout, dev)
    call('es_dconv_fwd_bf16', P(dyh), cout, P(wn), g, 1, cin, cout, P(dx), cin, 0, P(ws), nf, st_)
    wantx = torch.zeros(X, Y, Z, cin, device=dev)
    wantx[::2, ::2, ::2] = (dyh.float() @ wb.t()).view(X // 2, Y // 2, Z // 2, cin)
    assert float((dx - wantx.view(n_in, cin)).abs().max() / wantx.abs().max()) < 2e-5
    dw = torch.empty(1, cin, cout, device=dev)
    call('es_dconv_wgrad_bf16', P(xh), cin, P(dyh), cout, g, 0, cin, cout, P(dw), 0, st_)
    wantw = xs.t() @ dyh.float()
    assert float((dw[0] - wantw).abs().max() / wantw.abs().max()) < 2e-5


@pytest.mark.gpu
def test_fpn_output_convolution_dense_engine_vs_image_map_kernels():
    """The FPN's 3x3 output convolution at the occupancy configuration's size (10 views x 120 x 160 pixels, 256 -> 256, bias) through
    engine.conv: the dense engine on a flat grid (address arithmetic, bias pre-filled, weight gradient with its rows sliced over
    ~27 workgroups per tile) against the 9-wide image-map kernels -- forward, data gradient, weight and bias gradients through the tape.
    Tolerance 2e-5 of the largest magnitude (same bf16 operands, f32 accumulation order only)."""
    from embodiedscan_amd import engine as E, hip
    from embodiedscan_amd.models.backbones.resnet2d import _Grid
    dev = torch.device('cuda:0')
    gen = torch.Generator().manual_seed(31)
    n_img, h, w_, cin, cout = 10, 120, 160, 256, 256
    grid = _Grid(n_img, h, w_, dev)
    n = n_img * h * w_
    xd = torch.randn(n, cin, generator=gen).to(dev)
    wd = (torch.randn(9, cin, cout, generator=gen) / (9 * cin) ** 0.5).to(dev)
    bd = torch.randn(cout, generator=gen).to(dev)
    gy = torch.randn(n, cout, generator=gen).to(dev)
    assert int(hip.raw('es_dconv_wgrad_workspace_floats')(_geom(n_img, h, w_, 0, 3, 1, 1), 0, cin, cout)) > 0      # row slices
    res, prev_p, prev_d = {}, E.PRECISION[0], E.DENSE[0]
    E.PRECISION[0] = 'bf16'
    try:
        for dense_on in (True, False):
            E.DENSE[0] = dense_on
            E.begin_bind(('test_fpn', dense_on))
            try:
                w = E.Param(wd.clone(), torch.zeros_like(wd))
            finally:
                E.end_bind()
            b = E.Param(bd.clone(), torch.zeros_like(bd))
            x = E.Var(xd.clone())
            x.g = torch.ones_like(xd)
            E.TAPE.clear()
            E.new_grad_epoch()
            y = E.conv(x, w, None, None, n, bias=b, dense=(n_img, h, w_, 0, 3, 1, 1), maps=lambda: grid.conv_map(3, 1, 1)[:2])
            y.g = gy.clone()
            E.TAPE.backward()
            E.join_wgrad_streams()
            torch.cuda.synchronize()
            res[dense_on] = (y.d.clone(), x.g.clone(), w.g.clone(), b.g.clone())
            E.release(('test_fpn', dense_on))
    finally:
        E.PRECISION[0], E.DENSE[0] = prev_p, prev_d
        E.TAPE.clear()
    for a, b_, name in zip(res[True], res[False], ('y', 'dx', 'dw', 'db')):
        err = float((a - b_).abs().max() / b_.abs().max())
        assert err < 2e-5, (name, err)
