"""The halo-tile sparse convolution (embodiedscan_amd/csrc/halo.hip: es_halo_plan + es_spconv_halo_bf16) under the CDNA emulator of
tests/emu: the plan against a numpy restatement (sorted distinct source rows per 256-row tile, 16-bit positions), the convolution
against an f64 evaluation on the bf16-rounded operands -- local maps (the halo fits), ragged last tiles, two column tiles,
bias / accumulate, absent neighbours, inactive taps, and SCATTERED maps whose halo exceeds the 704 resident rows (paged path) --
under two thread schedules and with late LDS-DMA delivery; and against the gather kernel (es_spconv_fwd_bf16) on the same
operands.  TEST INFRASTRUCTURE: the product binds libes_hip.so only."""
import ctypes

import numpy as np
import pytest

from test_emu_kernels import P, _conv_ref, bf16_bits, bf16_round, emu  # noqa: F401  (the fixture)

K = 27


def _local_map(rng, n_out, n_in, fill, spread=40, dead_taps=()):
    """neighbours near the row's own index (what a Z-ordered set looks like): halo of a 256-row tile <= 256 + 2 * spread"""
    nbr = np.full((n_out, K), -1, np.int32)
    for k in range(K):
        if k in dead_taps:
            continue
        m = rng.random(n_out) < fill
        off = rng.integers(-spread, spread + 1, n_out)
        src = np.clip(np.arange(n_out) * n_in // max(n_out, 1) + off, 0, n_in - 1)
        nbr[m, k] = src[m]
    return nbr


def _plan(emu, nbr):
    n_out = nbr.shape[0]
    rows = emu.fns['es_halo_plan_rows'](n_out)
    tiles = rows // 256
    loc = np.full((rows, K), 0x1234, np.uint16)
    hrows = np.full((tiles, 256 * K), -7, np.int32)
    hcnt = np.full(tiles, -1, np.int32)
    emu('es_halo_plan', P(nbr), n_out, K, P(loc), P(hrows), P(hcnt), 0)
    return loc, hrows, hcnt


def _check_plan(nbr, loc, hrows, hcnt):
    n_out = nbr.shape[0]
    for t in range(len(hcnt)):
        blk = nbr[t * 256:(t + 1) * 256]
        want = np.unique(blk[blk >= 0])
        assert hcnt[t] == len(want)
        assert np.array_equal(hrows[t, :len(want)], want)                        # sorted, distinct
        lt = loc[t * 256:t * 256 + len(blk)]
        assert np.array_equal(lt == 0xFFFF, blk < 0)
        assert np.array_equal(want[lt[blk >= 0]], blk[blk >= 0])                 # position -> source row
    pad = loc[n_out:]
    assert (pad == 0xFFFF).all()                                                 # rows of the padding: absent everywhere


def _run(emu, rng, n_out, n_in, cin, cout, nbr, bias=False, accumulate=False, ldy_extra=0):
    x = rng.standard_normal((n_in, cin)).astype(np.float32)
    w = (rng.standard_normal((K, cin, cout)) / np.sqrt(K * cin)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32) if bias else None
    wt, wn = np.zeros((K, cout, cin), np.uint16), np.zeros((K, cin, cout), np.uint16)
    emu('es_cast_weight_bf16', P(w), K, cin, cout, P(wn), P(wt), 0)
    xh = bf16_bits(x)
    loc, hrows, hcnt = _plan(emu, nbr)
    _check_plan(nbr, loc, hrows, hcnt)
    ldy = cout + ldy_extra
    y0 = rng.standard_normal((n_out, ldy)).astype(np.float32)
    y = y0.copy()
    emu('es_spconv_halo_bf16', P(xh), cin, P(wt), P(loc), P(hrows), P(hcnt), n_out, n_in, K, cin, cout, P(b), P(y), ldy, int(accumulate), 0, 0)
    want = _conv_ref(bf16_round(x), bf16_round(w), nbr, b)
    if accumulate:
        want = want + y0[:, :cout]
    scale = max(np.abs(want).max(), 1e-6)
    err = np.abs(y[:, :cout] - want).max() / scale
    assert err < 2e-6, err
    if ldy_extra:
        assert np.array_equal(y[:, cout:], y0[:, cout:])                         # columns past Cout untouched
    return xh, wt, b, y, int(hcnt.max())


@pytest.mark.parametrize('lazy', [0, 1])
def test_halo_convolution_matches_f64_and_the_gather_kernel(emu, lazy):
    rng = np.random.default_rng(31 + lazy)
    emu.lib.es_emu_set_dma_mode(lazy)
    try:
        cases = [  # n_out, n_in, cin, cout, fill, dead taps, bias, accumulate, ldy_extra
            (300, 300, 64, 128, 0.6, (), True, False, 0),          # ragged second tile, bias
            (515, 400, 128, 128, 0.35, (0, 5, 26), False, True, 8),   # two chunks, three tiles, inactive taps, accumulate, padded rows
            (256, 256, 64, 256, 0.9, (), False, False, 0),          # two column tiles
        ] if not lazy else [(300, 300, 64, 128, 0.6, (3,), True, True, 0)]
        for n_out, n_in, cin, cout, fill, dead, bias, acc, ext in cases:
            nbr = _local_map(rng, n_out, n_in, fill, dead_taps=dead)
            xh, wt, b, y, umax = _run(emu, rng, n_out, n_in, cin, cout, nbr, bias, acc, ext)
            assert umax <= 704
            if not acc and not ext:                               # the gather kernel on the same operands: same products, another order
                y2 = np.zeros((n_out, cout), np.float32)
                emu('es_spconv_fwd_bf16', P(xh), 1, cin, P(wt), P(nbr), n_out, n_in, K, cin, cout, P(b), P(y2), cout, 0, 0)
                assert np.abs(y - y2).max() <= 2e-6 * max(np.abs(y2).max(), 1e-6)
    finally:
        emu.lib.es_emu_set_dma_mode(0)


def test_halo_pages_when_the_halo_exceeds_the_resident_rows(emu):
    """a scattered map (rows in hash order, not Z order): ~1 500 distinct source rows per tile -> three pages of 704"""
    rng = np.random.default_rng(77)
    n_out, n_in, cin, cout = 300, 3000, 64, 128
    nbr = np.full((n_out, K), -1, np.int32)
    m = rng.random((n_out, K)) < 0.3
    nbr[m] = rng.integers(0, n_in, int(m.sum()))
    *_, umax = _run(emu, rng, n_out, n_in, cin, cout, nbr, bias=True)
    assert umax > 2 * 704                                          # really paged


@pytest.mark.parametrize('lazy', [0, 1])
def test_halo_one_tap_groups(emu, lazy):
    """a single active tap: every step is the first AND the last of its group (chunk / page) -- the pipeline's halo re-staging
    right behind the step that reads it; with two chunks, and with a scattered tap (pages)"""
    rng = np.random.default_rng(91 + lazy)
    emu.lib.es_emu_set_dma_mode(lazy)
    try:
        n_out, cin, cout = 300, 128, 128
        nbr = np.full((n_out, K), -1, np.int32)
        nbr[:, 13] = np.arange(n_out)
        _run(emu, rng, n_out, n_out, cin, cout, nbr)
        n_in = 4000
        nbr = np.full((n_out, K), -1, np.int32)
        nbr[:, 4] = rng.permutation(n_in)[:n_out]                # 256 distinct rows per tile... one page
        nbr[:, 9] = rng.permutation(n_in)[:n_out]
        nbr[:, 20] = rng.permutation(n_in)[:n_out]               # 3 x 256 > 704: two pages, three taps
        *_, umax = _run(emu, rng, n_out, n_in, cin, cout, nbr, bias=True)
        assert umax > 704
        nbr2 = np.full((n_out, K), -1, np.int32)
        nbr2[:, 7] = nbr[:, 4]
        _run(emu, rng, n_out, n_in, 192, cout, nbr2)             # three chunks, one tap
    finally:
        emu.lib.es_emu_set_dma_mode(0)


def test_halo_persistent_workgroups_walk_several_tiles(emu):
    """8 workgroups for 10 x 2 tiles: every workgroup runs two or three tiles back to back (LDS reuse across tiles, the XCD-range
    tile numbering, two column tiles)"""
    rng = np.random.default_rng(123)
    emu('es_halo_set_option', 31, 8)
    try:
        n_out, n_in, cin, cout = 2400, 2400, 64, 256
        nbr = _local_map(rng, n_out, n_in, 0.3, dead_taps=(1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12))
        _run(emu, rng, n_out, n_in, cin, cout, nbr, bias=True)
        assert any('k_spconv_halo' in ln for ln in emu.launches())
    finally:
        emu('es_halo_set_option', 31, 0)


def test_halo_mirrored_taps_run_the_data_gradient_on_the_forward_plan(emu):
    """mirror = 1: tap k gathers through the plan's column 26 - k (the data gradient of a stride-1 convolution on one set:
    inv[i][k] == nbr[i][26 - k]) -- equals the convolution over the column-flipped map, incl. the tile's active-tap mask"""
    rng = np.random.default_rng(8)
    n, cin, cout = 300, 64, 128
    nbr = _local_map(rng, n, n, 0.5, dead_taps=(0, 1, 2, 20))
    x = rng.standard_normal((n, cin)).astype(np.float32)
    w = (rng.standard_normal((K, cin, cout)) / np.sqrt(K * cin)).astype(np.float32)
    wt, wn = np.zeros((K, cout, cin), np.uint16), np.zeros((K, cin, cout), np.uint16)
    emu('es_cast_weight_bf16', P(w), K, cin, cout, P(wn), P(wt), 0)
    xh = bf16_bits(x)
    loc, hrows, hcnt = _plan(emu, nbr)
    y = np.zeros((n, cout), np.float32)
    emu('es_spconv_halo_bf16', P(xh), cin, P(wt), P(loc), P(hrows), P(hcnt), n, n, K, cin, cout, 0, P(y), cout, 0, 1, 0)
    want = _conv_ref(bf16_round(x), bf16_round(w), np.ascontiguousarray(nbr[:, ::-1]), None)
    assert np.abs(y - want).max() < 2e-6 * np.abs(want).max()


def test_halo_empty_rows_and_support_rule(emu):
    rng = np.random.default_rng(5)
    n_out, n_in, cin, cout = 256, 100, 64, 128
    nbr = np.full((n_out, K), -1, np.int32)                        # no neighbour at all: zeros (+ bias)
    _run(emu, rng, n_out, n_in, cin, cout, nbr, bias=True)
    sup = emu.fns['es_spconv_halo_supported']
    assert sup(370000, 370000, 128, 27, 128, 128) == 1
    assert sup(2000, 2000, 256, 27, 256, 256) == 0                # under-filled: the tap-split gather kernels' case
    assert sup(370000, 370000, 128, 27, 128, 64) == 0             # 64 output channels
    assert sup(370000, 370000, 32, 27, 32, 128) == 0              # 32 reduction channels
    assert sup(370000, 370000, 128, 8, 128, 128) == 0
    x = np.zeros((4, 64), np.uint16)
    rc = emu.fns['es_spconv_halo_bf16'](P(x), 64, P(x), P(x), P(x), P(x), 4, 4, 27, 32, 128, 0, P(x), 128, 0, 0, 0)
    assert rc == -4
