"""Three kernel paths written at the end of round 3 and first run on hardware in round 4 (profiles/r4a_experimental.txt: all green).
(1) and (2) are ON by default since; (3) is a run-time option (es_set_option 10 = 3) that measured no faster than the
two-buffer kernel on the 64 .. 256-channel sparse layers (profiles/r4a_sweep.txt).
(1) the generative transposed convolution of the head's up-blocks as ONE launch per direction (engine.GEN_FUSED / es_gen_transpose_fwd_bf16 / es_gen_transpose_dgrad_bf16) against the
eight per-tap launches: forward bit-identical (same products, same order), data gradient equal to 1e-6 relative (the taps are
summed in one accumulator chain instead of eight read-modify-write passes), weight gradients untouched.
(2) the 128 x 128 weight-gradient tile with LDS-DMA staging and transposed LDS reads (es_set_option key 14,
k_spconv_wgrad_bf16_tr) against the register-transposing tile: same pairs, same chunk order, so bit-identical weight gradients --
IF ds_read_b64_tr_b16 has the lane mapping the kernel assumes (tools/probes/tr_read.hip prints the real one)."""
import os

import pytest
import torch

pytestmark = [pytest.mark.gpu]


@pytest.mark.parametrize('n,cin,cout', [(740, 1024, 512), (5920, 512, 256), (47360, 256, 128), (333, 64, 96)])
def test_fused_generative_transpose_matches_the_per_tap_launches(n, cin, cout):
    from embodiedscan_amd import engine as E
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(n)
    x = torch.randn(n, cin, generator=g).to(dev)
    w = (torch.randn(8, cin, cout, generator=g) / cin ** 0.5).to(dev)
    dy = torch.randn(n * 8, cout, generator=g).to(dev)
    res = {}
    E.PRECISION[0] = 'bf16'
    try:
        for fused in (False, True):
            E.GEN_FUSED[0] = fused
            E.TAPE.clear()
            E.WEIGHT_VERSION[0] += 1
            xv, wp = E.Var(x.clone()), E.Param(w.clone(), torch.zeros_like(w))
            y = E.gen_conv_transpose(xv, wp)
            y.g = dy.clone()
            E.TAPE.backward()
            torch.cuda.synchronize()
            res[fused] = (y.d.clone(), xv.g.clone(), wp.g.clone())
    finally:
        E.GEN_FUSED[0] = True
        E.PRECISION[0] = 'f32'
    assert torch.equal(res[True][0], res[False][0]), float((res[True][0] - res[False][0]).abs().max())
    e = float((res[True][1] - res[False][1]).norm() / res[False][1].norm())
    assert e < 1e-6, e
    assert torch.equal(res[True][2], res[False][2])
    print(f'fused generative transpose n={n} {cin}->{cout}: forward identical, data gradient rel-L2 {e:.1e}, weight gradient identical')


@pytest.mark.parametrize('cin,cout', [(128, 128), (256, 128), (128, 256)])
def test_transposed_read_weight_gradient_tile_matches_the_register_transposing_tile(cin, cout):
    from embodiedscan_amd import hip, sparse
    from embodiedscan_amd.hip import call, P
    dev = torch.device('cuda:0')
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(cin + cout)
    pts = [(torch.rand(30000, 3, generator=g) * 4 - 2).to(dev), (torch.rand(15000, 3, generator=g) * 4 - 2).to(dev)]
    cs, _ = sparse.voxelize(pts, 0.04)
    nbr = cs.kernel_map(cs, 3)
    n, K = cs.n, 27
    xh = torch.randn(n, cin, generator=g).to(dev).bfloat16().contiguous()
    dyh = torch.randn(n, cout, generator=g).to(dev).bfloat16().contiguous()
    need = int(hip.raw('es_spconv_wgrad_workspace_floats')(1, P(xh), 1, cin, P(dyh), 1, cout, n, n, K, cin, cout))
    ws = torch.empty(max(need, 1), device=dev)
    out = {}
    try:
        for mode in (0, 1):
            hip.raw('es_set_option')(14, mode)
            dw = torch.zeros(K, cin, cout, device=dev)
            call('es_spconv_wgrad_bf16_src', P(xh), 1, cin, P(dyh), 1, cout, P(nbr), n, n, K, cin, cout, P(dw), 0, P(ws), ws.numel(), st)
            torch.cuda.synchronize()
            out[mode] = dw
    finally:
        hip.raw('es_set_option')(14, 1)
    d = float((out[1] - out[0]).abs().max())
    assert torch.equal(out[1], out[0]), (cin, cout, d, float(out[0].abs().max()))
    print(f'transposed-read weight-gradient tile {cin}->{cout} on {n} voxels: identical to the register-transposing tile')


def test_three_buffer_dma_ring_matches_the_register_staged_kernel():
    """(3) k_spconv_bf16_dma<*, 1, 3> (es_set_option key 10 value 3): a ring of three LDS buffers with two chunks of DMA in
    flight (counted vmcnt + raw s_barrier) -- same products, same order as the register-staged kernel, so bit-identical."""
    from embodiedscan_amd import hip, sparse
    from embodiedscan_amd.hip import call, P
    dev = torch.device('cuda:0')
    st = torch.cuda.current_stream().cuda_stream
    opt = hip.raw('es_set_option')
    g = torch.Generator().manual_seed(9)
    pts = [(torch.rand(30000, 3, generator=g) * 4 - 2).to(dev), (torch.rand(15000, 3, generator=g) * 4 - 2).to(dev)]
    cs, _ = sparse.voxelize(pts, 0.04)
    small, _ = sparse.voxelize([(torch.rand(1500, 3, generator=g) * 4 - 2).to(dev)], 0.04)
    try:
        for S in (cs, small):
            nbr = S.kernel_map(S, 3)
            n, K = S.n, 27
            for cin, cout in ((32, 64), (128, 128), (256, 256), (96, 128)):
                xh = torch.randn(n, cin, generator=g).to(dev).bfloat16().contiguous()
                w = (torch.randn(K, cin, cout, generator=g) / (K * cin) ** 0.5).to(dev)
                wt = torch.empty((K, cout, cin), dtype=torch.bfloat16, device=dev)
                wn = torch.empty((K, cin, cout), dtype=torch.bfloat16, device=dev)
                call('es_cast_weight_bf16', P(w), K, cin, cout, P(wn), P(wt), st)
                bias = torch.randn(cout, generator=g).to(dev)
                nf = int(hip.raw('es_spconv_split_workspace_floats')(n, K, cin, cout))
                outs = {}
                for mode in (0, 3):
                    opt(10, mode)
                    opt(11, 0)
                    y = torch.empty(n, cout, device=dev)
                    if nf:
                        ws = torch.zeros(nf, device=dev)          # (head = tile tickets: zero on entry, left zero)
                        call('es_spconv_fwd_bf16_ws', P(xh), 1, cin, P(wt), P(nbr), n, n, K, cin, cout, P(bias), P(y), cout, 0, P(ws), nf, st)
                    else:
                        call('es_spconv_fwd_bf16', P(xh), 1, cin, P(wt), P(nbr), n, n, K, cin, cout, P(bias), P(y), cout, 0, st)
                    torch.cuda.synchronize()
                    outs[mode] = y
                assert torch.equal(outs[3], outs[0]), (n, cin, cout, float((outs[3] - outs[0]).abs().max()))
    finally:
        opt(10, 2)
        opt(11, 768)
    print('three-buffer DMA ring: identical to the register-staged kernel on 8 shapes')
