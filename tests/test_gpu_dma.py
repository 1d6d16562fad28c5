"""The LDS-DMA variant of the fast bf16 conv kernel (k_spconv_bf16_dma, es_set_option key 10) against the register-staged
ping-pong kernel it replaces: same bf16 products, same (tap, channel-chunk) accumulation order, so every output must be
IDENTICAL bit for bit -- 27-tap sparse maps (forward and transposed / data-gradient direction), strided maps, the identity map,
under-filled launches that split their tap list through the workspace, and the fused-epilogue / bf16-row modes of the image
backbone -- for both chunk sizes (32 and 64 channels) and both column-tile widths (C_out % 128 == 0 and 64).  One case is also
held to an f64 evaluation of the bf16-rounded operands (1e-6 of the output scale)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _case(dev, n, seed):
    from embodiedscan_amd import sparse
    g = torch.Generator().manual_seed(seed)
    pts = [(torch.rand(n, 3, generator=g) * 4 - 2).to(dev), (torch.rand(n // 2, 3, generator=g) * 4 - 2).to(dev)]
    cs, _ = sparse.voxelize(pts, 0.04)
    return cs


def test_dma_kernel_is_bit_identical_to_the_register_staged_kernel():
    from embodiedscan_amd import hip
    from embodiedscan_amd.hip import call, P
    dev = torch.device('cuda:0')
    st = torch.cuda.current_stream().cuda_stream
    opt = hip.raw('es_set_option')
    g = torch.Generator().manual_seed(5)
    cs = _case(dev, 30000, 1)
    small = _case(dev, 1500, 2)                       # few row tiles: the tap list is split over gridDim.z
    down = cs.strided(2)
    maps = [('3x3x3', cs.kernel_map(cs, 3), cs.n, cs.n), ('3x3x3 transposed', cs.inverse_map(cs, 3), cs.n, cs.n),
            ('stride 2', cs.kernel_map(down, 3), down.n, cs.n), ('stride 2 transposed', cs.inverse_map(down, 3), cs.n, down.n),
            ('3x3x3 under-filled', small.kernel_map(small, 3), small.n, small.n), ('identity', None, cs.n, cs.n)]
    n_checked = 0
    try:
        for name, nbr, n_out, n_in in maps:
            K = 1 if nbr is None else nbr.shape[1]
            for cin, cout in ((32, 64), (64, 128), (128, 128), (256, 256), (128, 64), (64, 192), (96, 128)):
                x = torch.randn(n_in, cin, generator=g).to(dev)
                xh = x.bfloat16().contiguous()
                w = (torch.randn(K, cin, cout, generator=g) / (K * cin) ** 0.5).to(dev)
                wt = torch.empty((K, cout, cin), dtype=torch.bfloat16, device=dev)
                wn = torch.empty((K, cin, cout), dtype=torch.bfloat16, device=dev)
                call('es_cast_weight_bf16', P(w), K, cin, cout, P(wn), P(wt), st)
                bias = torch.randn(cout, generator=g).to(dev)
                scale, shift = (torch.rand(cout, generator=g) + 0.5).to(dev), torch.randn(cout, generator=g).to(dev)
                res = torch.randn(n_out, cout, generator=g).to(dev)
                resh = res.bfloat16().contiguous()
                y0 = torch.randn(n_out, cout, generator=g).to(dev)
                nf = int(hip.raw('es_spconv_split_workspace_floats')(n_out, K, cin, cout))
                outs = {}
                for mode in (0, 1, 2):
                    opt(10, mode)
                    opt(11, 0)                       # every layer width (the default keeps the DMA kernel for >= 768 channels)
                    opt(3, 0)                        # K = 1: keep the launch on the conv kernels (not the row GEMM)
                    o = []
                    y = torch.empty(n_out, cout, device=dev)
                    call('es_spconv_fwd_bf16', P(xh), 1, cin, P(wt), P(nbr) if nbr is not None else 0, n_out, n_in, K, cin, cout,
                         P(bias), P(y), cout, 0, st)
                    o.append(y)
                    y = y0.clone()                   # accumulate into a strided output
                    wide = torch.zeros(n_out, 2 * cout, device=dev)
                    wide[:, cout:] = y
                    call('es_spconv_fwd_bf16', P(xh), 1, cin, P(wt), P(nbr) if nbr is not None else 0, n_out, n_in, K, cin, cout,
                         0, wide.data_ptr() + 4 * cout, 2 * cout, 1, st)
                    o.append(wide)
                    if nf:                           # deterministic tap split through the workspace: reduction by a second launch
                        for fold in (0, 1):          # (option 16 = 0, default) and by the tile's last workgroup (16 = 1) -- bit-identical
                            opt(16, fold)
                            ws = torch.zeros(nf, device=dev)          # (head = tile tickets: zero on entry, left zero)
                            y = torch.empty(n_out, cout, device=dev)
                            call('es_spconv_fwd_bf16_ws', P(xh), 1, cin, P(wt), P(nbr), n_out, n_in, K, cin, cout, P(bias), P(y), cout, 0,
                                 P(ws), nf, st)
                            if fold:
                                call('es_spconv_fwd_bf16_ws', P(xh), 1, cin, P(wt), P(nbr), n_out, n_in, K, cin, cout, P(bias), P(y), cout,
                                     0, P(ws), nf, st)       # the same workspace again: the tickets were left at zero
                                torch.cuda.synchronize()
                                assert int(ws[:1024].view(torch.int32).abs().max()) == 0
                            o.append(y)
                        opt(16, 0)
                        torch.cuda.synchronize()
                        assert torch.equal(o[-1], o[-2]), 'in-kernel split reduction differs from the two-launch reduction'
                    for act, r, rh, yh in ((1, res, 0, 0), (0, None, 0, 0), (3, res, 0, 0), (1, resh, 1, 1), (1, None, 0, 1),
                                           (3, resh, 1, 0)):
                        y = torch.empty(n_out, cout, device=dev, dtype=torch.bfloat16 if yh else torch.float32)
                        call('es_spconv_fwd_bf16_io', P(xh), 1, cin, P(wt), P(nbr) if nbr is not None else 0, n_out, n_in, K, cin, cout,
                             P(scale), P(shift) if act != 3 else 0, P(r) if r is not None else 0, rh, cout if r is not None else 0,
                             act, P(y), yh, cout, st)
                        o.append(y)
                    torch.cuda.synchronize()
                    outs[mode] = o
                for mode in (1, 2):
                    for i, (a, b) in enumerate(zip(outs[mode], outs[0])):
                        assert torch.equal(a, b), (name, cin, cout, 'chunk %d' % (32 * mode), 'output', i,
                                                   float((a.float() - b.float()).abs().max()))
                        n_checked += 1
                if name == '3x3x3' and (cin, cout) == (128, 128):
                    xb, wb = xh.double().cpu(), wn.double().cpu()
                    want = torch.zeros(n_out, cout, dtype=torch.float64)
                    nb = nbr.cpu().long()
                    for k in range(K):
                        rows = torch.nonzero(nb[:, k] >= 0).squeeze(1)
                        want[rows] += xb[nb[rows, k]] @ wb[k]
                    want += bias.double().cpu()
                    err = float((outs[2][0].double().cpu() - want).abs().max() / want.abs().max())
                    print(f'LDS-DMA kernel, 64-channel chunks, 3x3x3 128->128 on {n_out} voxels: vs f64 on the bf16 operands {err:.1e}')
                    assert err < 1e-6
    finally:
        opt(10, 2)
        opt(11, 768)
        opt(3, 1)
        opt(16, 0)
    print(f'LDS-DMA conv kernel: {n_checked} outputs identical to the register-staged kernel '
          f'({len(maps)} maps x 7 channel shapes x 2 chunk sizes x up to 9 launch modes)')
