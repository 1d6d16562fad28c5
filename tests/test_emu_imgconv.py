"""The 3x3 image convolution kernels (embodiedscan_amd/csrc/imgconv.hip: image rows in an LDS ring, weights in registers, fused
folded-BN / ReLU / gate epilogues) under the CDNA emulator of tests/emu: forward (stride 1 / 2, bf16 and f32 output rows) and gated
data gradient against f64 evaluations on the bf16-rounded operands, and against the map kernel path (es_spconv_fwd_bf16_io on
es_image_map / es_inverse_map) they replace -- 16 / 32 / 64 channels, ragged widths, several bands per image; two thread schedules,
late LDS-DMA delivery.  TEST INFRASTRUCTURE: the product binds libes_hip.so only."""
import numpy as np
import pytest

from test_emu_kernels import P, bf16_bits, bf16_round, emu  # noqa: F401  (the fixture)


def _fwd_ref(xb, wb, n_img, H, W, C, S):
    """f64 conv on channels-last rows: xb (n_img*H*W, C) bf16-rounded, wb (9, Cin, Cout) bf16-rounded -> (n_img*Ho*Wo, C)"""
    Ho, Wo = H // S, W // S
    x = np.zeros((n_img, H + 2, W + 2, C))
    x[:, 1:H + 1, 1:W + 1] = xb.reshape(n_img, H, W, C)
    y = np.zeros((n_img, Ho, Wo, C))
    for ty in range(3):
        for tx in range(3):
            y += x[:, ty:ty + S * Ho:S, tx:tx + S * Wo:S] @ wb[ty * 3 + tx].astype(np.float64)
    return y.reshape(-1, C)


CASES = [  # n_img, H, W (input grid), C, stride, workgroups aimed for
    (2, 5, 7, 16, 1, 2),
    (1, 9, 70, 16, 1, 3),          # WP = 128, three bands
    (2, 6, 33, 32, 1, 4),
    (1, 7, 20, 64, 1, 2),
    (2, 10, 14, 32, 2, 4),         # stride 2
]


@pytest.mark.parametrize('lazy', [0, 1])
def test_image_convolution_forward(emu, lazy):
    rng = np.random.default_rng(3 + lazy)
    emu.lib.es_emu_set_dma_mode(lazy)
    try:
        for n_img, H, W, C, S, wgs in (CASES if not lazy else CASES[1:3] + CASES[4:5]):
            assert emu.fns['es_img_conv3_supported'](n_img, H, W, C, S, 0) == 1
            emu('es_img_conv_set_option', 51, wgs)
            n, Ho, Wo = n_img * H * W, H // S, W // S
            n_o = n_img * Ho * Wo
            x = rng.standard_normal((n, C)).astype(np.float32)
            w = (rng.standard_normal((9, C, C)) / np.sqrt(9 * C)).astype(np.float32)
            scale, shift = (0.5 + rng.random(C)).astype(np.float32), rng.standard_normal(C).astype(np.float32)
            wt, wn = np.zeros((9, C, C), np.uint16), np.zeros((9, C, C), np.uint16)
            emu('es_cast_weight_bf16', P(w), 9, C, C, P(wn), P(wt), 0)
            xh = bf16_bits(x)
            conv = _fwd_ref(bf16_round(x), bf16_round(w), n_img, H, W, C, S)
            want = np.maximum(conv * scale + shift, 0)
            yf = np.full((n_o, C), np.nan, np.float32)
            emu('es_img_conv3_bf16', P(xh), C, P(wt), n_img, H, W, C, S, 0, P(scale), P(shift), 0, 0, 1, P(yf), 0, C, 0)
            err = np.abs(yf - want).max() / np.abs(want).max()
            assert err < 2e-6, (n_img, H, W, C, S, err)
            yh = np.zeros((n_o, C), np.uint16)
            emu('es_img_conv3_bf16', P(xh), C, P(wt), n_img, H, W, C, S, 0, P(scale), P(shift), 0, 0, 1, P(yh), 1, C, 0)
            assert np.array_equal(yh, bf16_bits(yf))                        # bf16 rows = the f32 result rounded
            # the map kernel it replaces, same operands (bf16 in, bf16 out)
            nbr = np.zeros((n_o, 9), np.int32)
            emu('es_image_map', n_img, H, W, Ho, Wo, 3, 3, S, 1, P(nbr), 0)
            y2 = np.zeros((n_o, C), np.uint16)
            emu('es_spconv_fwd_bf16_io', P(xh), 1, C, P(wt), P(nbr), n_o, n, 9, C, C, P(scale), P(shift), 0, 0, 0, 1, P(y2), 1, C, 0)
            d = np.abs(yh.astype(np.int32) - y2.astype(np.int32))
            assert d.max() <= 1 and (d > 0).mean() < 0.02, (d.max(), (d > 0).mean())      # (another summation order: a bf16 ulp here and there)
    finally:
        emu.lib.es_emu_set_dma_mode(0)
        emu('es_img_conv_set_option', 51, 1024)


def test_image_convolution_gated_data_gradient(emu):
    rng = np.random.default_rng(9)
    for n_img, H, W, C, wgs in ((2, 5, 7, 32, 2), (1, 9, 30, 64, 3), (1, 6, 50, 32, 1)):
        assert emu.fns['es_img_conv3_supported'](n_img, H, W, C, 1, 1) == 1
        emu('es_img_conv_set_option', 51, wgs)
        n = n_img * H * W
        gy = rng.standard_normal((n, C)).astype(np.float32)
        act = rng.standard_normal((n, C)).astype(np.float32)                 # the layer input's activation (its sign gates)
        w = (rng.standard_normal((9, C, C)) / np.sqrt(9 * C)).astype(np.float32)
        scale = (0.5 + rng.random(C)).astype(np.float32)
        wt, wn = np.zeros((9, C, C), np.uint16), np.zeros((9, C, C), np.uint16)
        emu('es_cast_weight_bf16', P(w), 9, C, C, P(wn), P(wt), 0)
        ah = bf16_bits(act)
        # dX[y][x][ci] = sum_t sum_co gy[y - ty + 1][x - tx + 1][co] W[t][ci][co]
        g = np.zeros((n_img, H + 2, W + 2, C))
        g[:, 1:H + 1, 1:W + 1] = bf16_round(gy).reshape(n_img, H, W, C)
        wb = bf16_round(w).astype(np.float64)
        dx = np.zeros((n_img, H, W, C))
        for ty in range(3):
            for tx in range(3):
                dx += g[:, 2 - ty:2 - ty + H, 2 - tx:2 - tx + W] @ wb[ty * 3 + tx].T
        want = np.where(bf16_round(act) > 0, dx.reshape(n, C) * scale, 0.0)
        out = np.full((n, C), np.nan, np.float32)
        emu('es_img_conv3_bf16', P(gy), C, P(wn), n_img, H, W, C, 1, 1, P(scale), 0, P(ah), C, 3, P(out), 0, C, 0)
        err = np.abs(out - want).max() / np.abs(want).max()
        assert err < 2e-6, (n_img, H, W, C, err)
        # the map kernel it replaces (gated data gradient through the inverse map)
        nbr, inv = np.zeros((n, 9), np.int32), np.zeros((n, 9), np.int32)
        emu('es_image_map', n_img, H, W, H, W, 3, 3, 1, 1, P(nbr), 0)
        emu('es_inverse_map', P(nbr), n, 9, n, P(inv), 0)
        out2 = np.zeros((n, C), np.float32)
        emu('es_spconv_fwd_bf16_io', P(gy), 0, C, P(wn), P(inv), n, n, 9, C, C, P(scale), 0, P(ah), 1, C, 3, P(out2), 0, C, 0)
        assert np.abs(out - out2).max() <= 3e-6 * np.abs(out2).max()
    emu('es_img_conv_set_option', 51, 1024)


def test_image_convolution_support_rule(emu):
    sup = emu.fns['es_img_conv3_supported']
    assert sup(80, 120, 120, 16, 1, 0) == 1 and sup(80, 60, 60, 32, 1, 0) == 1 and sup(80, 30, 30, 64, 1, 0) == 1
    assert sup(80, 120, 120, 32, 2, 0) == 1 and sup(80, 60, 60, 64, 2, 0) == 0        # (64 channels, stride 2: the map kernel is faster)
    assert sup(80, 60, 60, 32, 1, 1) == 1 and sup(80, 120, 120, 32, 2, 1) == 0        # no strided data gradient
    assert sup(80, 15, 15, 128, 1, 0) == 0 and sup(80, 120, 160, 32, 1, 0) == 0 and sup(80, 240, 240, 16, 2, 0) == 0


def test_stem_and_max_pool_in_one_launch(emu):
    """es_stem_pool_fwd (7x7 s2 p3 conv + frozen BN + ReLU + MaxPool2d(3, 2, 1), bf16 rows) against the two launches it replaces
    (es_stem_conv_fwd, es_maxpool_fwd_h on the 3x3 s2 p1 image map): bit-identical, on image sizes that are no multiple of any tile"""
    rng = np.random.default_rng(12)
    for n_img, H, W, C in ((2, 37, 50, 16), (1, 64, 30, 32), (3, 9, 11, 16), (1, 33, 41, 64)):
        Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        Hp, Wp = (Ho - 1) // 2 + 1, (Wo - 1) // 2 + 1
        x = rng.standard_normal((n_img, H, W, 3)).astype(np.float32)
        w = (rng.standard_normal((49, 3, C)) / 12).astype(np.float32)
        scale, shift = (rng.random(C) + 0.5).astype(np.float32), rng.standard_normal(C).astype(np.float32)
        y = np.zeros((n_img * Ho * Wo, C), np.float32)
        emu('es_stem_conv_fwd', P(x), P(w), P(scale), P(shift), n_img, H, W, C, P(y), 0)
        nbr = np.zeros((n_img * Hp * Wp, 9), np.int32)
        emu('es_image_map', n_img, Ho, Wo, Hp, Wp, 3, 3, 2, 1, P(nbr), 0)
        want = np.zeros((n_img * Hp * Wp, C), np.uint16)
        emu('es_maxpool_fwd_h', P(y), C, P(nbr), n_img * Hp * Wp, 9, C, P(want), 0)
        got = np.full((n_img * Hp * Wp, C), 0xffff, np.uint16)
        emu('es_stem_pool_fwd', P(x), P(w), P(scale), P(shift), n_img, H, W, C, P(got), 0)
        assert np.array_equal(got, want), (n_img, H, W, C, int((got != want).sum()))
