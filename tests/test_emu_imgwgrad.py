"""The 3x3 image weight-gradient kernel (embodiedscan_amd/csrc/imgwgrad.hip: image rows in an LDS ring, all nine taps per workgroup,
transposed LDS reads) under the CDNA emulator of tests/emu, against an f64 evaluation of the convolution's weight gradient on the
bf16-rounded operands and against the map kernel (es_spconv_wgrad_bf16_src on es_image_map) -- several images, ragged widths
(pad pixels), bands of output rows per workgroup (partial tensors through the workspace), strided rows, accumulation; two thread
schedules, late LDS-DMA delivery.  TEST INFRASTRUCTURE: the product binds libes_hip.so only."""
import numpy as np
import pytest

from test_emu_kernels import P, bf16_bits, bf16_round, emu  # noqa: F401  (the fixture)


def _ref(xb, gyb, n_img, H, W, C, S=1):
    """dW[t][ci][co] in f64: xb (n_img*H*W, C) on the input grid, gyb (n_img*(H/S)*(W/S), C) on the output grid, already bf16-rounded"""
    Ho, Wo = H // S, W // S
    x = np.zeros((n_img, H + 2, W + 2, C))
    x[:, 1:H + 1, 1:W + 1] = xb.reshape(n_img, H, W, C)
    g = gyb.reshape(n_img, Ho, Wo, C).astype(np.float64)
    dw = np.zeros((9, C, C))
    for ty in range(3):
        for tx in range(3):
            xs = x[:, ty:ty + S * Ho:S, tx:tx + S * Wo:S]
            dw[ty * 3 + tx] = np.einsum('nhwi,nhwo->io', xs, g)
    return dw


CASES = [  # n_img, H, W (input grid), C, workgroups aimed for (option 41 / 42), ld extra, accumulate, stride
    (2, 5, 7, 32, 2, 0, 0, 1),        # one band per image, WP = 64
    (1, 9, 70, 32, 4, 8, 1, 1),       # four bands (3 + 3 + 3 rows), WP = 128, strided rows, accumulate
    (2, 6, 33, 64, 6, 0, 0, 1),       # C = 64, three bands per image, WP = 64
    (1, 4, 20, 64, 1, 16, 1, 1),      # C = 64, WP = 32, one workgroup: writes dW directly (accumulating)
    (2, 10, 14, 32, 4, 0, 0, 2),      # stride 2: 5 x 7 outputs per image, two bands
    (1, 12, 72, 64, 3, 8, 1, 2),      # stride 2, C = 64, 36 outputs per row (WP = 64), three bands, strided rows, accumulate
]


@pytest.mark.parametrize('lazy', [0, 1])
def test_image_weight_gradient_matches_f64_and_the_map_kernel(emu, lazy):
    rng = np.random.default_rng(17 + lazy)
    emu.lib.es_emu_set_dma_mode(lazy)
    try:
        for n_img, H, W, C, wgs, ext, acc, S in (CASES if not lazy else CASES[1:3] + CASES[4:]):
            n, n_o = n_img * H * W, n_img * (H // S) * (W // S)
            x = rng.standard_normal((n, C + ext)).astype(np.float32)
            gy = rng.standard_normal((n_o, C + ext)).astype(np.float32)
            xh = bf16_bits(x)
            emu('es_img_wgrad_set_option', 41 if C == 32 else 42, wgs)
            nf = emu.fns['es_img_wgrad9_workspace_floats'](n_img, H, W, C, S)
            assert nf > 0 and nf % (9 * C * C) == 0
            ws = np.full(nf, np.nan, np.float32)
            dw0 = rng.standard_normal((9, C, C)).astype(np.float32)
            dw = dw0.copy()
            emu('es_img_wgrad9_bf16', P(xh), C + ext, P(gy), C + ext, n_img, H, W, C, S, P(dw), acc, P(ws), nf, 0)
            want = _ref(bf16_round(x)[:, :C], bf16_round(gy)[:, :C], n_img, H, W, C, S) + (dw0 if acc else 0)
            err = np.abs(dw - want).max() / np.abs(want).max()
            assert err < 2e-6, (n_img, H, W, C, err)
            if not ext and not acc:                   # the map kernel on the same operands (it rounds dY the same way)
                nbr = np.zeros((n_o, 9), np.int32)
                emu('es_image_map', n_img, H, W, H // S, W // S, 3, 3, S, 1, P(nbr), 0)
                dw2 = np.zeros((9, C, C), np.float32)
                emu('es_spconv_wgrad_bf16_src', P(xh), 1, C, P(gy), 0, C, P(nbr), n_o, n, 9, C, C, P(dw2), 0, 0, 0, 0)
                assert np.abs(dw - dw2).max() <= 3e-6 * np.abs(dw2).max()
    finally:
        emu.lib.es_emu_set_dma_mode(0)
        emu('es_img_wgrad_set_option', 41, 400)
        emu('es_img_wgrad_set_option', 42, 160)


def test_image_weight_gradient_support_rule(emu):
    wsf = emu.fns['es_img_wgrad9_workspace_floats']
    assert wsf(80, 120, 120, 32, 1) > 0 and wsf(80, 60, 60, 64, 1) > 0 and wsf(80, 120, 120, 32, 2) > 0 and wsf(80, 60, 60, 64, 2) > 0
    assert wsf(80, 120, 160, 32, 1) == 0       # wider than the LDS ring takes
    assert wsf(80, 30, 30, 128, 1) == 0        # 128 channels: the 128 x 128 tile's case
    assert wsf(80, 240, 240, 16, 1) == 0
    assert wsf(80, 15, 15, 32, 2) == 0         # odd grid under stride 2
    x = np.zeros((64, 64), np.float32)
    assert emu.fns['es_img_wgrad9_bf16'](P(x), 32, P(x), 32, 1, 4, 200, 32, 1, P(x), 0, P(x), 1 << 20, 0) == -4
    assert emu.fns['es_img_wgrad9_bf16'](P(x), 32, P(x), 32, 4, 8, 8, 32, 1, P(x), 0, 0, 0, 0) == -5       # several workgroups, no workspace


@pytest.mark.parametrize('lazy', [0, 1])
def test_rows_weight_gradient_of_1x1_layers(emu, lazy):
    """dW[Cin][Cout] = X^T dY on contiguous rows (es_rows_wgrad1_bf16): every (Cin tile, Cout tile) combination the plan picks, ragged last
    step, several slices, strided rows, accumulation -- against f64 on the bf16-rounded operands"""
    rng = np.random.default_rng(41 + lazy)
    emu.lib.es_emu_set_dma_mode(lazy)
    emu('es_img_wgrad_set_option', 43, 0)                 # (every width from 4 096 rows: the shipped rule takes < 256 channels only from 500 000)
    try:
        cases = [(4500, 32, 128, 0, 0), (4200, 128, 32, 8, 1), (4100, 64, 256, 0, 0), (4300, 256, 64, 0, 1), (4096, 64, 64, 16, 0),
                 (4400, 32, 64, 0, 0), (4200, 128, 128, 0, 0), (4150, 192, 32, 0, 0), (4130, 128, 320, 8, 1)]     # (last: one 128 x 320 tile)
        for n, cin, cout, ext, acc in (cases if not lazy else cases[:3] + cases[-1:]):
            nf = emu.fns['es_rows_wgrad1_workspace_floats'](n, cin, cout)
            assert nf > 0 and nf % (cin * cout) == 0, (n, cin, cout, nf)
            x = rng.standard_normal((n, cin + ext)).astype(np.float32)
            gy = rng.standard_normal((n, cout + ext)).astype(np.float32)
            xh = bf16_bits(x)
            ws = np.full(nf, np.nan, np.float32)
            dw0 = rng.standard_normal((cin, cout)).astype(np.float32)
            dw = dw0.copy()
            emu('es_rows_wgrad1_bf16', P(xh), cin + ext, P(gy), cout + ext, n, cin, cout, P(dw), acc, P(ws), nf, 0)
            want = bf16_round(x)[:, :cin].astype(np.float64).T @ bf16_round(gy)[:, :cout].astype(np.float64) + (dw0 if acc else 0)
            err = np.abs(dw - want).max() / np.abs(want).max()
            assert err < 2e-6, (n, cin, cout, err)
        wsf = emu.fns['es_rows_wgrad1_workspace_floats']
        assert wsf(1000, 64, 64) == 0 and wsf(100000, 16, 64) == 0 and wsf(100000, 32, 32) == 0 and wsf(100000, 1024, 256) == 0
        assert wsf(288000, 32, 128) > 0 and wsf(18000, 128, 512) > 0
        emu('es_img_wgrad_set_option', 43, 500000)
        assert wsf(288000, 32, 128) == 0 and wsf(864000, 32, 128) > 0 and wsf(72000, 256, 64) > 0
        assert wsf(60224, 128, 320) > 0 and wsf(60224, 64, 320) == 0
        emu('es_img_wgrad_set_option', 44, 0)
        assert wsf(60224, 128, 320) == 0
        emu('es_img_wgrad_set_option', 44, 1)
    finally:
        emu.lib.es_emu_set_dma_mode(0)
        emu('es_img_wgrad_set_option', 43, 500000)
