"""The dense-volume convolution engine (embodiedscan_amd/csrc/dconv.hip: address-arithmetic implicit GEMM, 256 / 320-row tiles,
LDS-DMA staging, transposed LDS reads in the weight gradient) under the CDNA emulator of tests/emu: forward, stride-1 data
gradient and weight gradient against f64 evaluations of nn.Conv3d's arithmetic on the bf16-rounded operands -- ragged row
tiles, every border case of a small volume, stride 2, batch > 1, both loop orders, forced slice counts (partial tiles through
the workspace), accumulation into the output -- under two thread schedules and with late LDS-DMA delivery.  TEST
INFRASTRUCTURE: the product binds libes_hip.so only."""
import numpy as np
import pytest

from test_emu_kernels import P, bf16_bits, bf16_round, emu  # noqa: F401  (the fixture)


def _conv3d_ref(xb, wb, B, X, Y, Z, ks, st, pad):
    """f64 nn.Conv3d on channels-last rows: xb (B*X*Y*Z, Cin), wb (K, Cin, Cout), taps ordered (kx, ky, kz)"""
    o = lambda d: (d + 2 * pad - ks) // st + 1
    Xo, Yo, Zo = o(X), o(Y), o(Z)
    cin, cout = wb.shape[1], wb.shape[2]
    xv = np.zeros((B, X + 2 * pad, Y + 2 * pad, Z + 2 * pad, cin))
    xv[:, pad:pad + X, pad:pad + Y, pad:pad + Z] = xb.reshape(B, X, Y, Z, cin)
    y = np.zeros((B, Xo, Yo, Zo, cout))
    for kx in range(ks):
        for ky in range(ks):
            for kz in range(ks):
                sl = xv[:, kx:kx + st * Xo:st, ky:ky + st * Yo:st, kz:kz + st * Zo:st]
                y += sl @ wb[(kx * ks + ky) * ks + kz].astype(np.float64)
    return y.reshape(-1, cout), (Xo, Yo, Zo)


def _geom(B, X, Y, Z, ks, st, pad):
    return np.array([B, X, Y, Z, ks, st, pad], np.int32)


CASES = [  # B, X, Y, Z, stride, Cin, Cout
    (1, 7, 6, 5, 1, 64, 256),        # 210 rows: one ragged tile, every border case
    (2, 9, 8, 5, 1, 128, 256),       # 720 rows: several tiles, batch boundary inside a tile
    (1, 10, 8, 6, 2, 64, 256),       # stride 2: 5 x 4 x 3 output voxels
    (1, 9, 7, 5, 1, 64, 128),        # 128 output channels: the 128-column tile (the neck's out blocks)
]


@pytest.mark.parametrize('lazy', [0, 1])
def test_dense_forward_and_data_gradient(emu, lazy):
    rng = np.random.default_rng(21 + lazy)
    emu.lib.es_emu_set_dma_mode(lazy)
    try:
        for ci, (B, X, Y, Z, st, cin, cout) in enumerate(CASES if not lazy else CASES[:2] + CASES[3:]):
            g = _geom(B, X, Y, Z, 3, st, 1)
            assert emu.fns['es_dconv_supported'](P(g), 0, cin, cout) == 1
            x = rng.standard_normal((B * X * Y * Z, cin)).astype(np.float32)
            w = (rng.standard_normal((27, cin, cout)) / np.sqrt(27 * cin)).astype(np.float32)
            wt, wn = np.zeros((27, cout, cin), np.uint16), np.zeros((27, cin, cout), np.uint16)
            emu('es_cast_weight_bf16', P(w), 27, cin, cout, P(wn), P(wt), 0)
            want, (Xo, Yo, Zo) = _conv3d_ref(bf16_round(x), bf16_round(w), B, X, Y, Z, 3, st, 1)
            M = B * Xo * Yo * Zo
            scale = np.abs(want).max()
            xh = bf16_bits(x)
            variants = [(0, 1, 0), (256, 0, 0), (320, 1, 3)] if not lazy else [(320, 1, 0), (256, 1, 2)]
            for rows, order, split in variants:
                emu('es_dconv_set_option', 20, rows)
                emu('es_dconv_set_option', 21, order)
                emu('es_dconv_set_option', 22, split)
                nf = int(emu.fns['es_dconv_workspace_floats'](P(g), 0, cin, cout))
                assert (nf > 0) == (split > 1) or split == 0
                ws = np.full(max(nf, 4), np.nan, np.float32)
                y = np.full((M, cout), np.nan, np.float32)
                emu.launches()
                emu('es_dconv_fwd_bf16', P(xh), cin, P(wt), P(g), 0, cin, cout, P(y), cout, 0, P(ws), nf, 0)
                ran = emu.launches()
                assert any('k_dconv' in k for k in ran), ran
                err = np.abs(y - want).max() / scale
                assert err < 2e-6, (ci, rows, order, split, err)
            # accumulate into Y
            y2 = np.ones((M, cout), np.float32)
            emu('es_dconv_fwd_bf16', P(xh), cin, P(wt), P(g), 0, cin, cout, P(y2), cout, 1, P(ws), nf, 0)
            assert np.abs(y2 - 1 - want).max() / scale < 2e-6
            if st != 1 or cin % 256:
                continue
            # data gradient of the stride-1 convolution: dX = conv(dY, flipped taps, W^T); reference = the adjoint identity
            dy = rng.standard_normal((M, cout)).astype(np.float32)
            dyb = bf16_round(dy)
            wf = bf16_round(w)[::-1].transpose(0, 2, 1)                  # (K, Cout, Cin), taps mirrored
            want_dx, _ = _conv3d_ref(dyb, wf, B, X, Y, Z, 3, 1, 1)
            emu('es_dconv_set_option', 20, 0); emu('es_dconv_set_option', 21, 0); emu('es_dconv_set_option', 22, 0)
            nf = int(emu.fns['es_dconv_workspace_floats'](P(g), 1, cin, cout))
            ws = np.zeros(max(nf, 4), np.float32)
            dx = np.full((B * X * Y * Z, cin), np.nan, np.float32)
            dyh = bf16_bits(dy)
            emu('es_dconv_fwd_bf16', P(dyh), cout, P(wn), P(g), 1, cin, cout, P(dx), cin, 0, P(ws), nf, 0)
            assert np.abs(dx - want_dx).max() / np.abs(want_dx).max() < 2e-6
    finally:
        emu.lib.es_emu_set_dma_mode(0)
        for k in (20, 22):
            emu('es_dconv_set_option', k, 0)
        emu('es_dconv_set_option', 21, 0)


@pytest.mark.parametrize('lazy', [0, 1])
def test_dense_weight_gradient(emu, lazy):
    rng = np.random.default_rng(31 + lazy)
    emu.lib.es_emu_set_dma_mode(lazy)
    try:
        for B, X, Y, Z, st in ((1, 6, 5, 5, 1), (2, 6, 6, 4, 2)) if not lazy else ((1, 5, 5, 4, 1),):
            cin = cout = 256
            g = _geom(B, X, Y, Z, 3, st, 1)
            assert emu.fns['es_dconv_supported'](P(g), 2, cin, cout) == 1
            o = lambda d: (d + 2 - 3) // st + 1
            Xo, Yo, Zo = o(X), o(Y), o(Z)
            M = B * Xo * Yo * Zo
            x = rng.standard_normal((B * X * Y * Z, cin)).astype(np.float32)
            dy = rng.standard_normal((M, cout)).astype(np.float32)
            xb, gb = bf16_round(x).astype(np.float64), bf16_round(dy).astype(np.float64)
            xv = np.zeros((B, X + 2, Y + 2, Z + 2, cin))
            xv[:, 1:1 + X, 1:1 + Y, 1:1 + Z] = xb.reshape(B, X, Y, Z, cin)
            want = np.zeros((27, cin, cout))
            for kx in range(3):
                for ky in range(3):
                    for kz in range(3):
                        sl = xv[:, kx:kx + st * Xo:st, ky:ky + st * Yo:st, kz:kz + st * Zo:st].reshape(M, cin)
                        want[(kx * 3 + ky) * 3 + kz] = sl.T @ gb
            dw = np.full((27, cin, cout), np.nan, np.float32)
            xh, dyh = bf16_bits(x), bf16_bits(dy)                        # (kept alive: P() of a temporary would dangle)
            emu.launches()
            emu('es_dconv_wgrad_bf16', P(xh), cin, P(dyh), cout, P(g), 0, cin, cout, P(dw), 0, 0)
            assert any('k_dconv_wgrad' in k for k in emu.launches())
            scale = np.abs(want).max()
            assert np.abs(dw - want).max() / scale < 2e-6, (B, X, Y, Z, st)
            dw2 = np.ones((27, cin, cout), np.float32)
            emu('es_dconv_wgrad_bf16', P(xh), cin, P(dyh), cout, P(g), 0, cin, cout, P(dw2), 1, 0)
            assert np.abs(dw2 - 1 - want).max() / scale < 2e-6
    finally:
        emu.lib.es_emu_set_dma_mode(0)


from test_emu_product import emulated, _ListAsDict  # noqa: E402,F401  (the fixture that puts the product's host layer on the emulator)


def _launch_log():
    import ctypes
    import build as emu_build
    lib = ctypes.CDLL(emu_build.build())
    buf = ctypes.create_string_buffer(1 << 16)
    lib.es_emu_take_launch_log(buf, len(buf))
    return buf.value.decode()


def test_engine_dense_path_equals_map_path_through_the_tape(emulated, monkeypatch):
    """engine.conv(dense=..., maps=...) in bf16 mode: forward, data gradient (accumulating into an existing gradient) and weight
    gradient through the tape on the dense engine against the same calls on the neighbour-map kernels (ES_DENSE off) -- the
    host-side dispatch of the occupancy neck (IndoorImVoxelNeck._conv3), stride 1 and 2, on the emulated library."""
    import torch
    from embodiedscan_amd import engine as E, hip
    from embodiedscan_amd.models.necks.imvoxel_neck import VolumeGrid
    dev = emulated
    monkeypatch.setitem(_ListAsDict(E.PRECISION), 0, 'bf16')
    gen = torch.Generator().manual_seed(4)
    for B, X, Y, Z, st, cin, cout in ((1, 6, 5, 4, 1, 256, 256), (1, 8, 6, 4, 2, 256, 256)):
        grid = VolumeGrid(B, X, Y, Z, dev)
        o = lambda d: (d + 2 - 3) // st + 1
        n_out = B * o(X) * o(Y) * o(Z)
        xd = torch.randn(B * X * Y * Z, cin, generator=gen)
        wd = torch.randn(27, cin, cout, generator=gen) / (27 * cin) ** 0.5
        gy = torch.randn(n_out, cout, generator=gen)
        res = {}
        for dense_on in (True, False):
            monkeypatch.setitem(_ListAsDict(E.DENSE), 0, dense_on)
            x = E.Var(xd.clone())
            x.g = torch.ones_like(xd)                                  # an existing gradient: the data-gradient launch accumulates
            w = E.Param(wd.clone(), torch.zeros_like(wd))
            w.bf_n, w.bf_t = torch.empty((27, cin, cout), dtype=torch.bfloat16), torch.empty((27, cout, cin), dtype=torch.bfloat16)
            hip.call('es_cast_weight_bf16', hip.P(w.d), 27, cin, cout, hip.P(w.bf_n), hip.P(w.bf_t), 0)
            w.bf_step = E.WEIGHT_VERSION[0]
            E.TAPE.clear()
            E.new_grad_epoch()
            _launch_log()
            y = E.conv(x, w, None, None, n_out, dense=(B, X, Y, Z, 3, st, 1), maps=lambda: grid.conv_map(3, st, 1)[:2])
            y.g = gy.clone()
            E.TAPE.backward()
            log = _launch_log()
            # dense on: forward + weight gradient (+ the stride-1 data gradient) on the dense engine, maps only for the strided dgrad
            assert ('k_dconv<' in log) == dense_on and ('k_dconv_wgrad' in log) == dense_on, log
            assert ('k_volume_map' in log) == (not dense_on or st != 1) or 'k_volume_map' not in log
            res[dense_on] = (y.d.clone(), x.g.clone(), w.g.clone())
        for a, b, name in zip(res[True], res[False], ('y', 'dx', 'dw')):
            err = float((a - b).abs().max() / b.abs().max())
            assert err < 2e-5, (name, st, err)


def _fine(t, B, X, Y, Z, C):
    """(B*2X*2Y*2Z, C) rows -> (8, B*X*Y*Z, C): class p = (px, py, pz) holds the rows of the voxels (2x+px, 2y+py, 2z+pz)"""
    v = t.reshape(B, X, 2, Y, 2, Z, 2, C)
    return np.stack([v[:, :, px, :, py, :, pz].reshape(-1, C) for px in range(2) for py in range(2) for pz in range(2)])


@pytest.mark.parametrize('lazy', [0, 1])
def test_dense_strided_data_gradient_and_transposed_convolution(emu, lazy):
    """the parity-class launches: data gradient of nn.Conv3d(k=3, s=2, p=1) (each input voxel through the 1 / 2 / 4 / 8 taps that
    reach it) and nn.ConvTranspose3d(k=2, s=2) forward written straight into dense order, plus the transposed convolution's data
    and weight gradients -- against f64 adjoints on the bf16-rounded operands"""
    rng = np.random.default_rng(41 + lazy)
    emu.lib.es_emu_set_dma_mode(lazy)
    try:
        for B, Xo, Yo, Zo, cin, cout in ((1, 5, 4, 3, 256, 64), (2, 4, 3, 3, 256, 128)) if not lazy else ((1, 4, 3, 2, 256, 64),):
            # ---- strided data gradient: input grid (2Xo, 2Yo, 2Zo), cin channels; output grid (Xo, Yo, Zo), cout channels
            X, Y, Z = 2 * Xo, 2 * Yo, 2 * Zo
            g = _geom(B, X, Y, Z, 3, 2, 1)
            assert emu.fns['es_dconv_supported'](P(g), 1, cin, cout) == 1
            w = (rng.standard_normal((27, cin, cout)) / np.sqrt(27 * cout)).astype(np.float32)
            wt, wn = np.zeros((27, cout, cin), np.uint16), np.zeros((27, cin, cout), np.uint16)
            emu('es_cast_weight_bf16', P(w), 27, cin, cout, P(wn), P(wt), 0)
            M = B * Xo * Yo * Zo
            dy = rng.standard_normal((M, cout)).astype(np.float32)
            dyh = bf16_bits(dy)
            dyb, wb = bf16_round(dy).astype(np.float64), bf16_round(w).astype(np.float64)
            dxp = np.zeros((B, X + 2, Y + 2, Z + 2, cin))
            for kx in range(3):
                for ky in range(3):
                    for kz in range(3):
                        dxp[:, kx:kx + 2 * Xo:2, ky:ky + 2 * Yo:2, kz:kz + 2 * Zo:2] += (dyb @ wb[(kx * 3 + ky) * 3 + kz].T).reshape(B, Xo, Yo, Zo, cin)
            want = dxp[:, 1:1 + X, 1:1 + Y, 1:1 + Z].reshape(-1, cin)
            for rows, split in ((0, 0), (256, 1), (320, 1), (0, 2), (256, 3)):      # split > 1: slices of every class, class-major partial rows
                emu('es_dconv_set_option', 20, rows)
                emu('es_dconv_set_option', 22, split)
                nf = int(emu.fns['es_dconv_workspace_floats'](P(g), 1, cin, cout))
                wsb = np.full(max(nf, 4), np.nan, np.float32)
                dx = np.full((B * X * Y * Z, cin), np.nan, np.float32)
                emu.launches()
                emu('es_dconv_fwd_bf16', P(dyh), cout, P(wn), P(g), 1, cin, cout, P(dx), cin, 0, P(wsb), nf, 0)
                ran = emu.launches()
                assert any('k_dconv<' in k for k in ran) and (any('k_dconv_reduce_cls' in k for k in ran) == (nf > 0))
                assert np.abs(dx - want).max() / np.abs(want).max() < 2e-6, ('strided dgrad', rows, split)
                dx2 = np.ones((B * X * Y * Z, cin), np.float32)
                emu('es_dconv_fwd_bf16', P(dyh), cout, P(wn), P(g), 1, cin, cout, P(dx2), cin, 1, P(wsb), nf, 0)
                assert np.abs(dx2 - 1 - want).max() / np.abs(want).max() < 2e-6, ('strided dgrad, accumulate', rows, split)
            emu('es_dconv_set_option', 20, 0)
            emu('es_dconv_set_option', 22, 0)
            # ---- transposed convolution (k = 2, s = 2): coarse grid (Xo, Yo, Zo) with ci_t channels -> fine grid with co_t channels
            ci_t, co_t = 256, 256
            gt = _geom(B, Xo, Yo, Zo, 2, 2, 0)
            assert emu.fns['es_dconv_supported'](P(gt), 3, ci_t, co_t) == 1
            x = rng.standard_normal((M, ci_t)).astype(np.float32)
            w8 = (rng.standard_normal((8, ci_t, co_t)) / np.sqrt(ci_t)).astype(np.float32)
            w8t, w8n = np.zeros((8, co_t, ci_t), np.uint16), np.zeros((8, ci_t, co_t), np.uint16)
            emu('es_cast_weight_bf16', P(w8), 8, ci_t, co_t, P(w8n), P(w8t), 0)
            xh = bf16_bits(x)
            xb, w8b = bf16_round(x).astype(np.float64), bf16_round(w8).astype(np.float64)
            y = np.full((8 * M, co_t), np.nan, np.float32)
            nf3 = int(emu.fns['es_dconv_workspace_floats'](P(gt), 3, ci_t, co_t))
            ws3 = np.full(max(nf3, 4), np.nan, np.float32)
            emu('es_dconv_fwd_bf16', P(xh), ci_t, P(w8t), P(gt), 3, ci_t, co_t, P(y), co_t, 0, P(ws3), nf3, 0)
            got = _fine(y, B, Xo, Yo, Zo, co_t)
            for p in range(8):
                wantp = xb @ w8b[p]
                assert np.abs(got[p] - wantp).max() / np.abs(wantp).max() < 2e-6, ('transposed fwd', p)
            if ci_t % 256 == 0 or True:
                # data gradient: dX[r] = sum_p dY[2r + p] W[p]^T  (N = ci_t must be a multiple of 256 for the dense engine)
                dyf = rng.standard_normal((8 * M, co_t)).astype(np.float32)
                dyfh = bf16_bits(dyf)
                cls = _fine(bf16_round(dyf).astype(np.float64), B, Xo, Yo, Zo, co_t)
                assert emu.fns['es_dconv_supported'](P(gt), 4, ci_t, co_t) == 1 and emu.fns['es_dconv_supported'](P(gt), 5, ci_t, co_t) == 1
                if True:
                    wantx = sum(cls[p] @ w8b[p].T for p in range(8))
                    dxc = np.full((M, ci_t), np.nan, np.float32)
                    nf = int(emu.fns['es_dconv_workspace_floats'](P(gt), 4, ci_t, co_t))
                    wsb = np.zeros(max(nf, 4), np.float32)
                    emu('es_dconv_fwd_bf16', P(dyfh), co_t, P(w8n), P(gt), 4, ci_t, co_t, P(dxc), ci_t, 0, P(wsb), nf, 0)
                    assert np.abs(dxc - wantx).max() / np.abs(wantx).max() < 2e-6, 'transposed dgrad'
                if emu.fns['es_dconv_supported'](P(gt), 5, ci_t, co_t) == 1:
                    wantw = np.stack([xb.T @ cls[p] for p in range(8)])
                    dw = np.full((8, ci_t, co_t), np.nan, np.float32)
                    emu('es_dconv_wgrad_bf16', P(xh), ci_t, P(dyfh), co_t, P(gt), 1, ci_t, co_t, P(dw), 0, 0)
                    assert np.abs(dw - wantw).max() / np.abs(wantw).max() < 2e-6, 'transposed wgrad'
    finally:
        emu.lib.es_emu_set_dma_mode(0)
        emu('es_dconv_set_option', 20, 0)
        emu('es_dconv_set_option', 22, 0)


def test_dense_pointwise_stride2_downsample(emu):
    """the neck's identity down-sample nn.Conv3d(k=1, s=2, p=0): forward (one tap, source 2 r), data gradient (parity classes:
    class 0 alone has a tap, the other seven write zeros -- or leave an accumulated gradient alone) and weight gradient"""
    rng = np.random.default_rng(77)
    for B, Xo, Yo, Zo, cin, cout in ((1, 5, 4, 3, 256, 256), (2, 3, 3, 2, 256, 512)):
        X, Y, Z = 2 * Xo, 2 * Yo, 2 * Zo
        g = _geom(B, X, Y, Z, 1, 2, 0)
        for mode in (0, 1, 2):
            assert emu.fns['es_dconv_supported'](P(g), mode, cin, cout) == 1, mode
        n_in, M = B * X * Y * Z, B * Xo * Yo * Zo
        x = rng.standard_normal((n_in, cin)).astype(np.float32)
        w = (rng.standard_normal((1, cin, cout)) / np.sqrt(cin)).astype(np.float32)
        wt, wn = np.zeros((1, cout, cin), np.uint16), np.zeros((1, cin, cout), np.uint16)
        emu('es_cast_weight_bf16', P(w), 1, cin, cout, P(wn), P(wt), 0)
        xh = bf16_bits(x)
        xb, wb = bf16_round(x).astype(np.float64), bf16_round(w).astype(np.float64)[0]
        xs = xb.reshape(B, X, Y, Z, cin)[:, ::2, ::2, ::2].reshape(M, cin)              # the sampled voxels
        # forward
        nf = int(emu.fns['es_dconv_workspace_floats'](P(g), 0, cin, cout))
        wsb = np.full(max(nf, 4), np.nan, np.float32)
        y = np.full((M, cout), np.nan, np.float32)
        emu('es_dconv_fwd_bf16', P(xh), cin, P(wt), P(g), 0, cin, cout, P(y), cout, 0, P(wsb), nf, 0)
        want = xs @ wb
        assert np.abs(y - want).max() / np.abs(want).max() < 2e-6
        # data gradient: dX[2 r] = dY[r] W^T, every other voxel 0
        dy = rng.standard_normal((M, cout)).astype(np.float32)
        dyh = bf16_bits(dy)
        dyb = bf16_round(dy).astype(np.float64)
        wantx = np.zeros((B, X, Y, Z, cin))
        wantx[:, ::2, ::2, ::2] = (dyb @ wb.T).reshape(B, Xo, Yo, Zo, cin)
        wantx = wantx.reshape(n_in, cin)
        nf = int(emu.fns['es_dconv_workspace_floats'](P(g), 1, cin, cout))
        wsb = np.full(max(nf, 4), np.nan, np.float32)
        dx = np.full((n_in, cin), np.nan, np.float32)
        emu('es_dconv_fwd_bf16', P(dyh), cout, P(wn), P(g), 1, cin, cout, P(dx), cin, 0, P(wsb), nf, 0)
        assert np.abs(dx - wantx).max() / np.abs(wantx).max() < 2e-6
        dx2 = np.ones((n_in, cin), np.float32)
        emu('es_dconv_fwd_bf16', P(dyh), cout, P(wn), P(g), 1, cin, cout, P(dx2), cin, 1, P(wsb), nf, 0)
        assert np.abs(dx2 - 1 - wantx).max() / np.abs(wantx).max() < 2e-6
        # weight gradient
        dw = np.full((1, cin, cout), np.nan, np.float32)
        emu('es_dconv_wgrad_bf16', P(xh), cin, P(dyh), cout, P(g), 0, cin, cout, P(dw), 0, 0)
        wantw = xs.T @ dyb
        assert np.abs(dw[0] - wantw).max() / np.abs(wantw).max() < 2e-6


def test_engine_dense_transposed_convolution_equals_the_generative_path(emulated, monkeypatch):
    """engine.conv_transpose_dense (parity-class launch, dense row order) against gather_rows(gen_conv_transpose(...), up_index) --
    the path it replaces in IndoorImVoxelNeck -- forward, data gradient and the 8 weight-gradient taps, through the tape"""
    import torch
    from embodiedscan_amd import engine as E, hip
    from embodiedscan_amd.models.necks.imvoxel_neck import VolumeGrid
    dev = emulated
    monkeypatch.setitem(_ListAsDict(E.PRECISION), 0, 'bf16')
    gen = torch.Generator().manual_seed(6)
    B, X, Y, Z, cin, cout = 1, 3, 4, 2, 256, 256
    geo = (B, X, Y, Z, 2, 2, 0)
    assert all(E.dense_ok(geo, m, cin, cout) for m in (3, 4, 5))
    grid = VolumeGrid(B, X, Y, Z, dev)
    n = B * X * Y * Z
    xd = torch.randn(n, cin, generator=gen)
    wd = torch.randn(8, cin, cout, generator=gen) / cin ** 0.5
    gy = torch.randn(8 * n, cout, generator=gen)
    res = {}
    for dense_on in (True, False):
        x = E.Var(xd.clone())
        w = E.Param(wd.clone(), torch.zeros_like(wd))
        w.bf_n, w.bf_t = torch.empty((8, cin, cout), dtype=torch.bfloat16), torch.empty((8, cout, cin), dtype=torch.bfloat16)
        hip.call('es_cast_weight_bf16', hip.P(w.d), 8, cin, cout, hip.P(w.bf_n), hip.P(w.bf_t), 0)
        w.bf_step = E.WEIGHT_VERSION[0]
        E.TAPE.clear()
        E.new_grad_epoch()
        y = E.conv_transpose_dense(x, w, geo) if dense_on else E.gather_rows(E.gen_conv_transpose(x, w), grid.up_index())
        y.g = gy.clone()
        E.TAPE.backward()
        res[dense_on] = (y.d.clone(), x.g.clone(), w.g.clone())
    for a, b, name in zip(res[True], res[False], ('y', 'dx', 'dw')):
        err = float((a - b).abs().max() / b.abs().max())
        assert err < 2e-5, (name, err)


def test_adamw_table_equals_flat_adamw_plus_weight_cast(emu):
    """es_adamw_table (AdamW + the bf16 copies of the kernels in one pass) against es_adamw_step followed by es_cast_weights_table:
    the same bits in the parameters, both moments and both copies -- ragged kernels (3 x 64 stem, 100 x 36), plain ranges between
    them, paramwise multipliers, clipping active"""
    import struct
    rng = np.random.default_rng(5)
    shapes = [(27, 3, 64), (1, 100, 36), (2, 128, 70), (8, 64, 64)]
    gaps = [5, 4100, 0, 12, 33]                                   # plain elements before / between / after the kernels
    n = sum(gaps) + sum(k * a * b for k, a, b in shapes)
    p0 = rng.standard_normal(n).astype(np.float32)
    g = rng.standard_normal(n).astype(np.float32)
    m0 = (0.1 * rng.standard_normal(n)).astype(np.float32)
    v0 = (0.01 * rng.random(n)).astype(np.float32)
    norm = np.array([np.sqrt((g.astype(np.float64) ** 2).sum())], np.float32)
    lr, wd, step, max_norm, gs = 1e-3, 1e-2, 7, 10.0, 0.5
    dbits = lambda x: struct.unpack('<q', struct.pack('<d', float(x)))[0]
    mult = [(1.0, 1.0), (0.1, 1.0), (1.0, 0.0)]
    # ---- reference: flat AdamW per multiplier range, then the cast table
    pr, mr, vr = p0.copy(), m0.copy(), v0.copy()
    rows, cast_rows, items, tiles, off = [], [], 0, 0, 0
    keep = []
    for i in range(len(shapes) + 1):
        lm, dm = mult[i % 3]
        if gaps[i]:
            rows.append([off, gaps[i], 0, 0, 0, 0, items, dbits(lm), dbits(dm)])
            items += (gaps[i] + 4095) // 4096
            sl = slice(off, off + gaps[i])
            a, b_, c, d = pr[sl].copy(), g[sl].copy(), mr[sl].copy(), vr[sl].copy()
            emu('es_adamw_step', P(a), P(b_), P(c), P(d), gaps[i], lr * lm, 0.9, 0.999, 1e-8, wd * dm, step, max_norm, P(norm), gs, 0)
            pr[sl], mr[sl], vr[sl] = a, c, d
            off += gaps[i]
        if i < len(shapes):
            K, A, B = shapes[i]
            cnt = K * A * B
            nat, tr = np.zeros(cnt, np.uint16), np.zeros(cnt, np.uint16)
            nat2, tr2 = np.zeros(cnt, np.uint16), np.zeros(cnt, np.uint16)
            keep.append((nat, tr, nat2, tr2, off, cnt))
            rows.append([off, K, A, B, nat.ctypes.data, tr.ctypes.data, items, dbits(lm), dbits(dm)])
            items += K * ((A + 63) // 64) * ((B + 63) // 64)
            sl = slice(off, off + cnt)
            a, b_, c, d = pr[sl].copy(), g[sl].copy(), mr[sl].copy(), vr[sl].copy()
            emu('es_adamw_step', P(a), P(b_), P(c), P(d), cnt, lr * lm, 0.9, 0.999, 1e-8, wd * dm, step, max_norm, P(norm), gs, 0)
            pr[sl], mr[sl], vr[sl] = a, c, d
            off += cnt
    assert off == n
    for nat, tr, nat2, tr2, o, cnt in keep:                       # cast table over the REFERENCE parameters (pointers into pr)
        K, A, B = shapes[len(cast_rows)]
        cast_rows.append([pr.ctypes.data + 4 * o, nat2.ctypes.data, tr2.ctypes.data, K, A, B, tiles])
        tiles += K * ((A + 63) // 64) * ((B + 63) // 64)
    ct = np.array(cast_rows, np.int64)
    emu('es_cast_weights_table', P(ct), len(cast_rows), tiles, 0)
    # ---- one pass
    p1, m1, v1 = p0.copy(), m0.copy(), v0.copy()
    t = np.array([[x - (1 << 64) if x >= (1 << 63) else x for x in r] for r in rows], np.int64)
    emu('es_adamw_table', P(p1), P(g), P(m1), P(v1), P(t), len(rows), items, lr, 0.9, 0.999, 1e-8, wd, step, max_norm, P(norm), gs, 0)
    assert np.array_equal(p1, pr) and np.array_equal(m1, mr) and np.array_equal(v1, vr)
    assert not np.array_equal(p1, p0)
    for nat, tr, nat2, tr2, o, cnt in keep:
        assert np.array_equal(nat, nat2) and np.array_equal(tr, tr2) and nat.any()


def _conv2d_ref(xb, wb, B, X, Y, st):
    """f64 nn.Conv2d(k=3, stride, pad=1) on channels-last rows: xb (B*X*Y, Cin), wb (9, Cin, Cout), taps ordered (kx, ky)"""
    Xo, Yo = (X + 2 - 3) // st + 1, (Y + 2 - 3) // st + 1
    cin, cout = wb.shape[1], wb.shape[2]
    xv = np.zeros((B, X + 2, Y + 2, cin))
    xv[:, 1:1 + X, 1:1 + Y] = xb.reshape(B, X, Y, cin)
    y = np.zeros((B, Xo, Yo, cout))
    for kx in range(3):
        for ky in range(3):
            y += xv[:, kx:kx + st * Xo:st, ky:ky + st * Yo:st] @ wb[kx * 3 + ky].astype(np.float64)
    return y.reshape(-1, cout), (Xo, Yo)


def test_dense_flat_grid_2d_convolution_with_bias_and_row_sliced_weight_gradient(emu):
    """geom[3] = 0: nn.Conv2d(k=3, p=1) on (B, X, Y) images by the same kernels -- forward onto rows pre-filled with the bias (also through the slice
    reduction of a forced split), stride-1 data gradient, stride-2 forward, and the weight gradient with its rows sliced over several
    workgroups per tile (partial tensors through the workspace) -- against f64 on the bf16-rounded operands"""
    rng = np.random.default_rng(99)
    for B, X, Y, st, cin, cout in ((2, 13, 11, 1, 256, 256), (1, 12, 10, 2, 256, 256)):
        g = _geom(B, X, Y, 0, 3, st, 1)
        assert emu.fns['es_dconv_supported'](P(g), 0, cin, cout) == 1 and emu.fns['es_dconv_supported'](P(g), 2, cin, cout) == 1
        assert emu.fns['es_dconv_supported'](P(g), 1, cin, cout) == (1 if st == 1 else 0)
        assert emu.fns['es_dconv_supported'](P(g), 3, cin, cout) == 0
        x = rng.standard_normal((B * X * Y, cin)).astype(np.float32)
        w = (rng.standard_normal((9, cin, cout)) / np.sqrt(9 * cin)).astype(np.float32)
        bias = rng.standard_normal(cout).astype(np.float32)
        wt, wn = np.zeros((9, cout, cin), np.uint16), np.zeros((9, cin, cout), np.uint16)
        emu('es_cast_weight_bf16', P(w), 9, cin, cout, P(wn), P(wt), 0)
        xh = bf16_bits(x)
        xb, wb = bf16_round(x).astype(np.float64), bf16_round(w).astype(np.float64)
        want, (Xo, Yo) = _conv2d_ref(xb, wb, B, X, Y, st)
        M = B * Xo * Yo
        for split in (0, 3):
            emu('es_dconv_set_option', 22, split)
            nf = int(emu.fns['es_dconv_workspace_floats'](P(g), 0, cin, cout))
            wsb = np.full(max(nf, 4), np.nan, np.float32)
            y = np.ascontiguousarray(np.broadcast_to(bias, (M, cout))).copy()      # the caller's bias: rows pre-filled, accumulate = 1
            emu('es_dconv_fwd_bf16', P(xh), cin, P(wt), P(g), 0, cin, cout, P(y), cout, 1, P(wsb), nf, 0)
            assert np.abs(y - (want + bias)).max() / np.abs(want).max() < 2e-6, ('flat fwd + bias', st, split)
            y3 = np.full((M, cout), np.nan, np.float32)
            emu('es_dconv_fwd_bf16', P(xh), cin, P(wt), P(g), 0, cin, cout, P(y3), cout, 0, P(wsb), nf, 0)
            assert np.abs(y3 - want).max() / np.abs(want).max() < 2e-6, ('flat fwd', st, split)
        emu('es_dconv_set_option', 22, 0)
        dy = rng.standard_normal((M, cout)).astype(np.float32)
        dyh = bf16_bits(dy)
        dyb = bf16_round(dy).astype(np.float64)
        if st == 1:                               # data gradient: the adjoint, tap by tap
            dxp = np.zeros((B, X + 2, Y + 2, cin))
            for kx in range(3):
                for ky in range(3):
                    dxp[:, kx:kx + X, ky:ky + Y] += (dyb @ wb[kx * 3 + ky].T).reshape(B, X, Y, cin)
            wantx = dxp[:, 1:1 + X, 1:1 + Y].reshape(-1, cin)
            nf = int(emu.fns['es_dconv_workspace_floats'](P(g), 1, cin, cout))
            wsb = np.full(max(nf, 4), np.nan, np.float32)
            dx = np.full((B * X * Y, cin), np.nan, np.float32)
            emu('es_dconv_fwd_bf16', P(dyh), cout, P(wn), P(g), 1, cin, cout, P(dx), cin, 0, P(wsb), nf, 0)
            assert np.abs(dx - wantx).max() / np.abs(wantx).max() < 2e-6, 'flat dgrad'
        # weight gradient: X gathered under the tap
        xv = np.zeros((B, X + 2, Y + 2, cin))
        xv[:, 1:1 + X, 1:1 + Y] = xb.reshape(B, X, Y, cin)
        wantw = np.stack([xv[:, kx:kx + st * Xo:st, ky:ky + st * Yo:st].reshape(M, cin).T @ dyb for kx in range(3) for ky in range(3)])
        for ws_split in (1, 2, 5):
            emu('es_dconv_set_option', 23, ws_split)
            nfw = int(emu.fns['es_dconv_wgrad_workspace_floats'](P(g), 0, cin, cout))
            assert (nfw > 0) == (ws_split > 1)
            wsw = np.full(max(nfw, 4), np.nan, np.float32)
            dw = np.full((9, cin, cout), np.nan, np.float32)
            emu.launches()
            emu('es_dconv_wgrad_ws_bf16', P(xh), cin, P(dyh), cout, P(g), 0, cin, cout, P(dw), 0, P(wsw), nfw, 0)
            ran = emu.launches()
            assert any('k_dconv_reduce' in k for k in ran) == (ws_split > 1)
            assert np.abs(dw - wantw).max() / np.abs(wantw).max() < 2e-6, ('flat wgrad', st, ws_split)
            dw2 = np.ones((9, cin, cout), np.float32)
            emu('es_dconv_wgrad_ws_bf16', P(xh), cin, P(dyh), cout, P(g), 0, cin, cout, P(dw2), 1, P(wsw), nfw, 0)
            assert np.abs(dw2 - 1 - wantw).max() / np.abs(wantw).max() < 2e-6, ('flat wgrad, accumulate', st, ws_split)
        emu('es_dconv_set_option', 23, 0)


def test_engine_fpn_output_convolution_dense_equals_image_map_path(emulated, monkeypatch):
    """engine.conv(bias=..., dense=(n_img, h, w, 0, 3, 1, 1), maps=...) -- the FPN's 3x3 output convolution (models/necks/fpn.py) -- on the
    dense engine (flat grid, bias pre-filled, row-sliced weight gradient) against the same call on the 9-wide image map: forward,
    data gradient, weight and bias gradients through the tape, on the emulated library (this also pins the tap order of the two paths)"""
    import torch
    from embodiedscan_amd import engine as E, hip
    from embodiedscan_amd.models.backbones.resnet2d import _Grid
    dev = emulated
    monkeypatch.setitem(_ListAsDict(E.PRECISION), 0, 'bf16')
    gen = torch.Generator().manual_seed(8)
    n_img, h, w_, cin, cout = 2, 9, 7, 256, 256
    grid = _Grid(n_img, h, w_, dev)
    n = n_img * h * w_
    xd = torch.randn(n, cin, generator=gen)
    wd = torch.randn(9, cin, cout, generator=gen) / (9 * cin) ** 0.5
    bd = torch.randn(cout, generator=gen)
    gy = torch.randn(n, cout, generator=gen)
    res = {}
    for dense_on in (True, False):
        monkeypatch.setitem(_ListAsDict(E.DENSE), 0, dense_on)
        x = E.Var(xd.clone())
        x.g = torch.ones_like(xd)
        w = E.Param(wd.clone(), torch.zeros_like(wd))
        b = E.Param(bd.clone(), torch.zeros_like(bd))
        w.bf_n, w.bf_t = torch.empty((9, cin, cout), dtype=torch.bfloat16), torch.empty((9, cout, cin), dtype=torch.bfloat16)
        hip.call('es_cast_weight_bf16', hip.P(w.d), 9, cin, cout, hip.P(w.bf_n), hip.P(w.bf_t), 0)
        w.bf_step = E.WEIGHT_VERSION[0]
        E.TAPE.clear()
        E.new_grad_epoch()
        _launch_log()
        y = E.conv(x, w, None, None, n, bias=b, dense=(n_img, h, w_, 0, 3, 1, 1), maps=lambda: grid.conv_map(3, 1, 1)[:2])
        y.g = gy.clone()
        E.TAPE.backward()
        log = _launch_log()
        assert ('k_dconv<' in log) == dense_on and ('k_dconv_wgrad' in log) == dense_on and ('k_image_map' in log) == (not dense_on), log
        res[dense_on] = (y.d.clone(), x.g.clone(), w.g.clone(), b.g.clone())
    for a, b_, name in zip(res[True], res[False], ('y', 'dx', 'dw', 'db')):
        err = float((a - b_).abs().max() / b_.abs().max())
        assert err < 2e-5, (name, err)
