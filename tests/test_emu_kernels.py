"""Kernel LOGIC on the CPU (tests/emu): the sources of embodiedscan_amd/csrc/{spconv,rowops}.hip compiled for x86 against an
emulation of the CDNA execution model (fibers per workgroup, wave rendezvous for shuffles / ballots / MFMA fragments, LDS-DMA,
the transposed LDS read) and driven through the SAME C ABI with host pointers.  What this pins without a GPU: tile indexing,
neighbour-map gathers, tap compaction and tap split, LDS swizzles, fragment layouts, epilogues, the norm pipelines' chunking and
their one-launch variants -- against numpy evaluations of the same arithmetic (bf16-rounded operands, f64 sums).  What it cannot
pin: anything about timing or the memory model (the emulator is more permissive than the hardware) -- the GPU suite does that.
The emulated library is test infrastructure: the product binds libes_hip.so only (embodiedscan_amd/hip.py) and has no CPU path."""
import ctypes
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests', 'emu'))


@pytest.fixture(scope='module', params=['descending', 'random'])
def emu(request):
    """the emulated library under two thread schedules: between two synchronisation points the threads of a workgroup run in
    descending or random order (tests/test_emu_product.py runs ascending) -- a kernel that is missing a barrier gives different
    results under them"""
    import build as emu_build
    from embodiedscan_amd import hip
    lib = ctypes.CDLL(emu_build.build())
    lib.es_emu_set_schedule.argtypes = [ctypes.c_int, ctypes.c_ulonglong]
    lib.es_emu_set_schedule(['ascending', 'descending', 'random'].index(request.param), 12345)
    fns = {}
    for name, (ret, at, _) in hip.PROTOS.items():
        f = getattr(lib, name, None)
        if f is not None:                      # (only the sources listed in tests/emu/build.py are part of the emulated library)
            f.restype, f.argtypes = ret, at
            fns[name] = f

    def call(name, *args):
        rc = fns[name](*args)
        assert rc == 0, (name, rc)

    def launches():
        """kernel expressions launched since the last call"""
        buf = ctypes.create_string_buffer(1 << 16)
        lib.es_emu_take_launch_log(buf, len(buf))
        return [ln.split(' grid=')[0] for ln in buf.value.decode().splitlines()]
    call.fns, call.launches, call.lib, call.schedule = fns, launches, lib, request.param
    return call


def test_the_emulator_catches_a_missing_wait_and_a_missing_barrier(emu):
    """the two failure classes the emulator exists to catch, on deliberately breakable kernels (tests/emu/selftest_kernels.cpp):
    LDS read before its LDS-DMA was waited for (seen with late DMA delivery), LDS exchange without a barrier (seen under at
    least one thread schedule); the correct variants pass under every mode"""
    lib = emu.lib
    rng = np.random.default_rng(5)
    src = rng.integers(0, 256, 3 * 4096).astype(np.uint8)
    src[src == 0xEE] = 1
    for lazy in (0, 1):
        lib.es_emu_set_dma_mode(lazy)
        for wait in (1, 0):
            dst = np.zeros_like(src)
            lib.es_emu_selftest_dma(ctypes.c_void_p(P(src)), ctypes.c_void_p(P(dst)), 3, wait)
            good = np.array_equal(dst, src)
            assert good == (wait == 1 or lazy == 0), (lazy, wait)          # the missing wait is invisible with eager delivery,
    lib.es_emu_set_dma_mode(0)                                             # caught with late delivery
    a = rng.integers(0, 1000, 2 * 256).astype(np.int32)
    want = np.concatenate([np.roll(a[i:i + 256], -1) + np.roll(a[i:i + 256], 1) for i in (0, 256)])
    for barrier in (1, 0):
        out = np.zeros_like(a)
        lib.es_emu_selftest_barrier(ctypes.c_void_p(P(a)), ctypes.c_void_p(P(out)), 2, barrier)
        if barrier:
            assert np.array_equal(out, want)
        else:
            assert not np.array_equal(out, want), emu.schedule             # every schedule runs SOME reader before its writer


def P(a):
    return a.ctypes.data if a is not None else 0


def bf16_round(x):
    """f32 -> nearest-even bf16, returned as f32 (the conversion every kernel uses: v_cvt_pk_bf16_f32)"""
    u = np.ascontiguousarray(x, np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32)


def bf16_bits(x):
    return (bf16_round(x).view(np.uint32) >> 16).astype(np.uint16)


def _map(rng, n_out, n_in, K, fill):
    nbr = np.full((n_out, K), -1, np.int32)
    m = rng.random((n_out, K)) < fill
    nbr[m] = rng.integers(0, n_in, int(m.sum()))
    return nbr


def _conv_ref(xb, wb, nbr, bias):
    """f64 evaluation on the bf16-rounded operands: xb (n_in, Cin), wb (K, Cin, Cout)"""
    n_out, K = nbr.shape
    y = np.zeros((n_out, wb.shape[2]), np.float64)
    for k in range(K):
        rows = np.flatnonzero(nbr[:, k] >= 0)
        if len(rows):
            y[rows] += xb[nbr[rows, k]].astype(np.float64) @ wb[k].astype(np.float64)
    return y + (bias.astype(np.float64) if bias is not None else 0)


def test_weight_cast_layouts(emu):
    rng = np.random.default_rng(0)
    K, A, B = 3, 40, 72
    w = rng.standard_normal((K, A, B)).astype(np.float32)
    nat = np.zeros((K, A, B), np.uint16)
    tr = np.zeros((K, B, A), np.uint16)
    emu('es_cast_weight_bf16', P(w), K, A, B, P(nat), P(tr), 0)
    assert np.array_equal(nat, bf16_bits(w)) and np.array_equal(tr, bf16_bits(w).transpose(0, 2, 1))


@pytest.mark.parametrize('dma', [0, 1, 2, 3])
def test_sparse_conv_forward_on_the_emulated_matrix_cores(emu, dma):
    """27-tap gather convolution, bf16 rows: the register-staged ping-pong kernel (option 10 = 0) and the LDS-DMA kernel with
    64-channel chunks (10 = 2, every width) against f64 on the rounded operands; a ragged last row tile, absent neighbours, a
    tap that no row of a tile uses (tap compaction), and an under-filled launch that splits its taps through the workspace"""
    rng = np.random.default_rng(1 + dma)
    emu('es_set_option', 10, dma)
    emu('es_set_option', 11, 0)
    try:
        for lazy, (n_out, n_in, K, cin, cout, fill) in [(lz, c) for lz in ((0, 1) if dma else (0,)) for c in ((300, 280, 27, 64, 128, 0.3), (130, 130, 27, 128, 64, 0.5), (77, 90, 8, 64, 64, 0.9))]:
            emu.lib.es_emu_set_dma_mode(lazy)                       # LDS-DMA delivered at issue / as late as the waits allow
            nbr = _map(rng, n_out, n_in, K, fill)
            nbr[:, 5 % K] = -1                                   # a tap nobody uses
            x = rng.standard_normal((n_in, cin)).astype(np.float32)
            w = (rng.standard_normal((K, cin, cout)) / np.sqrt(K * cin)).astype(np.float32)
            bias = rng.standard_normal(cout).astype(np.float32)
            xh = bf16_bits(x)
            wt = np.zeros((K, cout, cin), np.uint16)
            wn = np.zeros((K, cin, cout), np.uint16)
            emu('es_cast_weight_bf16', P(w), K, cin, cout, P(wn), P(wt), 0)
            want = _conv_ref(bf16_round(x), bf16_round(w), nbr, bias)
            scale = np.abs(want).max()
            y = np.full((n_out, cout), np.nan, np.float32)
            emu.launches()
            emu('es_spconv_fwd_bf16', P(xh), 1, cin, P(wt), P(nbr), n_out, n_in, K, cin, cout, P(bias), P(y), cout, 0, 0)
            ran = emu.launches()
            assert any(('k_spconv_bf16_dma' if dma else 'k_spconv_bf16_fast') in k for k in ran), ran
            err = np.abs(y - want).max() / scale
            assert err < 2e-6, (n_out, cin, cout, err)
            # f32 rows (converted while staged) through the same entry point: only the register-staged kernels take them
            y2 = np.full((n_out, cout), np.nan, np.float32)
            emu('es_spconv_fwd_bf16', P(x), 0, cin, P(wt), P(nbr), n_out, n_in, K, cin, cout, P(bias), P(y2), cout, 0, 0)
            assert np.abs(y2 - want).max() / scale < 2e-6
            # tap split through the workspace (+ the second launch that adds the slices): same sums
            nf = int(emu.fns['es_spconv_split_workspace_floats'](n_out, K, cin, cout))
            if nf:
                ws = np.zeros(nf, np.float32)
                y3 = np.full((n_out, cout), np.nan, np.float32)
                emu('es_spconv_fwd_bf16_ws', P(xh), 1, cin, P(wt), P(nbr), n_out, n_in, K, cin, cout, P(bias), P(y3), cout, 0, P(ws), nf, 0)
                assert np.abs(y3 - want).max() / scale < 2e-6
                assert not ws[:1024].view(np.int32).any()        # the tile tickets are left at zero
                # the weight-sharing workgroup order of a split launch (option 20) is a permutation of the same tiles: same bits
                emu('es_set_option', 20, 1)
                y4 = np.full((n_out, cout), np.nan, np.float32)
                emu('es_spconv_fwd_bf16_ws', P(xh), 1, cin, P(wt), P(nbr), n_out, n_in, K, cin, cout, P(bias), P(y4), cout, 0, P(ws), nf, 0)
                emu('es_set_option', 20, 0)
                assert np.array_equal(y4, y3)
    finally:
        emu.lib.es_emu_set_dma_mode(0)
        emu('es_set_option', 10, 2)
        emu('es_set_option', 11, 768)


def test_row_gemm_with_fused_epilogue(emu):
    """K = 1 on the identity map (every 1x1 convolution / Linear layer): both row-GEMM generations, affine + residual + ReLU"""
    rng = np.random.default_rng(7)
    n, cin, cout = 333, 64, 128
    x = rng.standard_normal((n, cin)).astype(np.float32)
    w = (rng.standard_normal((1, cin, cout)) / np.sqrt(cin)).astype(np.float32)
    wt = np.zeros((1, cout, cin), np.uint16)
    wn = np.zeros((1, cin, cout), np.uint16)
    emu('es_cast_weight_bf16', P(w), 1, cin, cout, P(wn), P(wt), 0)
    scale, shift = (rng.random(cout) + 0.5).astype(np.float32), rng.standard_normal(cout).astype(np.float32)
    res = rng.standard_normal((n, cout)).astype(np.float32)
    lin = bf16_round(x).astype(np.float64) @ bf16_round(w)[0].astype(np.float64)
    want = np.maximum(lin * scale + shift + res, 0)
    for gen2 in (1, 0):
        emu('es_set_option', 13, gen2)
        y = np.full((n, cout), np.nan, np.float32)
        emu('es_spconv_fwd_bf16_affine', P(x), cin, P(wt), 0, n, n, 1, cin, cout, P(scale), P(shift), P(res), cout, 1, P(y), cout, 0)
        assert np.abs(y - want).max() / np.abs(want).max() < 2e-6, gen2
    emu('es_set_option', 13, 1)


def test_row_gemm_whole_width_tile(emu):
    """the head's 128 -> 320 output GEMM (K = 1, identity map, bias, no second epilogue operand) as ONE 320-column tile (option 23):
    f32 and bf16 input rows, ragged last row tile; same bits as the 64-column tiles it replaces"""
    rng = np.random.default_rng(9)
    n, cin, cout = 16384 + 44, 128, 320          # (taken from 16 384 rows)
    x = rng.standard_normal((n, cin)).astype(np.float32)
    w = (rng.standard_normal((1, cin, cout)) / np.sqrt(cin)).astype(np.float32)
    bias = rng.standard_normal(cout).astype(np.float32)
    wt, wn = np.zeros((1, cout, cin), np.uint16), np.zeros((1, cin, cout), np.uint16)
    emu('es_cast_weight_bf16', P(w), 1, cin, cout, P(wn), P(wt), 0)
    want = bf16_round(x).astype(np.float64) @ bf16_round(w)[0].astype(np.float64) + bias
    xh = bf16_bits(x)
    got = {}
    for on in (1, 0):
        emu('es_set_option', 23, on)
        for half, rows in ((0, x), (1, xh)):
            y = np.full((n, cout), np.nan, np.float32)
            emu.launches()
            emu('es_spconv_fwd_bf16', P(rows), half, cin, P(wt), 0, n, n, 1, cin, cout, P(bias), P(y), cout, 0, 0)
            assert any('k_rowgemm2_bf16<320' in k for k in emu.launches()) == bool(on)
            assert np.abs(y - want).max() / np.abs(want).max() < 2e-6, (on, half)
            got[on, half] = y
    emu('es_set_option', 23, 1)
    assert np.array_equal(got[1, 0], got[0, 0]) and np.array_equal(got[1, 1], got[0, 1]) and np.array_equal(got[1, 0], got[1, 1])


def test_small_row_linear_kernel(emu):
    """K = 1 launches with few rows (the grounding decoder's linears) on the 64 x 64 whole-stage kernel (option 24): one and several
    256-channel stages, a last stage of 8 channels, ragged rows, bias, accumulation, strided input rows -- against f64 on the rounded
    operands and bit for bit against the 128-row kernel it replaces"""
    rng = np.random.default_rng(10)
    for n, cin, cout, ext, acc in ((300, 256, 256, 0, 0), (70, 64, 128, 4, 1), (130, 520, 64, 0, 0), (64, 2048, 256, 0, 1), (1, 256, 64, 0, 0)):
        xb = rng.standard_normal((n, cin + ext)).astype(np.float32)
        x = xb[:, :cin]
        w = (rng.standard_normal((1, cin, cout)) / np.sqrt(cin)).astype(np.float32)
        bias = rng.standard_normal(cout).astype(np.float32)
        wt, wn = np.zeros((1, cout, cin), np.uint16), np.zeros((1, cin, cout), np.uint16)
        emu('es_cast_weight_bf16', P(w), 1, cin, cout, P(wn), P(wt), 0)
        y0 = rng.standard_normal((n, cout)).astype(np.float32)
        want = bf16_round(x).astype(np.float64) @ bf16_round(w)[0].astype(np.float64) + bias + (y0 if acc else 0)
        got = {}
        for on in (256, 0):
            emu('es_set_option', 24, on)
            y = y0.copy() if acc else np.full((n, cout), np.nan, np.float32)
            emu.launches()
            emu('es_spconv_fwd_bf16', P(xb), 0, cin + ext, P(wt), 0, n, n, 1, cin, cout, P(bias), P(y), cout, acc, 0)
            assert any('k_lin_small' in k for k in emu.launches()) == bool(on)
            assert np.abs(y - want).max() / np.abs(want).max() < 2e-6, (n, cin, cout, on)
            got[on] = y
        emu('es_set_option', 24, 256)
        assert np.array_equal(got[256], got[0]), (n, cin, cout)


def test_small_row_linear_weight_gradient(emu):
    """dW = X^T dY of the few-row linear layers (es_spconv_wgrad_bf16, K = 1, identity map, f32 rows, <= 256 x 256 weights) on the 64 x 64 tile
    kernel: ragged last slice, several slices through the workspace and the single-slice form, strided rows, accumulation; the same shape
    with a map keeps the gather kernels"""
    rng = np.random.default_rng(13)
    for n, cin, cout, ext, acc in ((300, 256, 256, 0, 0), (700, 64, 128, 4, 1), (256, 128, 64, 0, 0), (5, 64, 64, 0, 1)):
        xb = rng.standard_normal((n, cin + ext)).astype(np.float32)
        gb = rng.standard_normal((n, cout + ext)).astype(np.float32)
        x, gy = xb[:, :cin], gb[:, :cout]
        want = bf16_round(x).astype(np.float64).T @ bf16_round(gy).astype(np.float64)
        dw0 = rng.standard_normal((1, cin, cout)).astype(np.float32)
        nf = int(emu.fns['es_spconv_wgrad_workspace_floats'](1, P(xb), 0, cin + ext, P(gb), 0, cout + ext, n, n, 1, cin, cout))
        assert nf == (-(-n // 256)) * cin * cout * (n > 256), (n, nf)
        ws = np.full(max(nf, 1), np.nan, np.float32)
        for use_ws in (1, 0):
            dw = dw0.copy()
            emu.launches()
            emu('es_spconv_wgrad_bf16', P(xb), cin + ext, P(gb), cout + ext, 0, n, n, 1, cin, cout, P(dw), acc, P(ws) if use_ws else 0, nf if use_ws else 0, 0)
            assert any('k_lin_wgrad_small' in k for k in emu.launches())
            err = np.abs(dw[0] - (want + (dw0[0] if acc else 0))).max() / np.abs(want).max()
            assert err < 2e-6, (n, cin, cout, use_ws, err)
        ident = np.arange(n, dtype=np.int32)[:, None].copy()
        dw = dw0.copy()
        emu.launches()
        emu('es_spconv_wgrad_bf16', P(xb), cin + ext, P(gb), cout + ext, P(ident), n, n, 1, cin, cout, P(dw), acc, P(ws), nf, 0)
        assert not any('k_lin_wgrad_small' in k for k in emu.launches())
        assert np.abs(dw[0] - (want + (dw0[0] if acc else 0))).max() / np.abs(want).max() < 2e-6


def test_expansion_convolution_stream_kernel(emu):
    """the image backbone's 1x1 C -> 4 C layers on bf16 rows (frozen BN (+ bf16 residual) (+ ReLU)) on the register-resident kernel (option 25)
    against the row GEMM it replaces: 16 / 32 / 64 input channels, ragged last tile, more tiles than persistent waves, strided rows"""
    rng = np.random.default_rng(21)
    for n, cin, with_res, act, ext, wgs in ((333, 16, 1, 1, 0, 1024), (1000, 32, 0, 0, 8, 3), (517, 64, 1, 1, 0, 5), (16, 64, 1, 0, 0, 1024)):
        cout = 4 * cin
        x = bf16_bits(rng.standard_normal((n, cin + ext)).astype(np.float32))
        res = bf16_bits(rng.standard_normal((n, cout + ext)).astype(np.float32))
        w = (rng.standard_normal((1, cin, cout)) / np.sqrt(cin)).astype(np.float32)
        wt, wn = np.zeros((1, cout, cin), np.uint16), np.zeros((1, cin, cout), np.uint16)
        emu('es_cast_weight_bf16', P(w), 1, cin, cout, P(wn), P(wt), 0)
        scale, shift = (rng.random(cout) + 0.5).astype(np.float32), rng.standard_normal(cout).astype(np.float32)
        got = {}
        for on in (1, 0):
            emu('es_set_option', 25, on)
            emu('es_set_option', 26, wgs)
            y = np.full((n, cout + ext), 0x7fc0, np.uint16)
            emu.launches()
            emu('es_spconv_fwd_bf16_io', P(x), 1, cin + ext, P(wt), 0, n, n, 1, cin, cout, P(scale), P(shift), P(res) if with_res else 0, 1, cout + ext,
                act, P(y), 1, cout + ext, 0)
            assert any('k_expand_bf16' in k for k in emu.launches()) == bool(on)
            got[on] = y
        emu('es_set_option', 25, 65536)
        emu('es_set_option', 26, 1024)
        assert np.array_equal(got[1], got[0]), (n, cin, int((got[1] != got[0]).sum()))
        assert (got[1][:, cout:] == 0x7fc0).all()


def test_weight_gradient_tiles(emu):
    """dW[k] = X[nbr[:, k]]^T dY through the bf16 weight-gradient kernels (64 x 64 tile, 128 x 128 tile, and the LDS-DMA +
    transposed-read tile whose lane mapping was probed on the GPU) -- row slices through the workspace included"""
    rng = np.random.default_rng(11)
    for n_out, n_in, K, cin, cout, tr in ((700, 650, 27, 64, 64, 0), (900, 900, 8, 128, 128, 0), (900, 900, 8, 128, 128, 1),
                                          (900, 900, 8, 128, 128, 2), (900, 900, 8, 128, 128, 3), (900, 900, 8, 128, 128, 4)):
        emu.lib.es_emu_set_dma_mode(1 if tr in (2, 4) else 0)       # the transposed-read tile stages by LDS-DMA: also with late delivery
        tr = {0: 0, 1: 1, 2: 1, 3: 2, 4: 2}[tr]                    # option 14: 1 = 32 pairs per chunk, 2 = 64 (round 6)
        emu('es_set_option', 14, tr)
        nbr = _map(rng, n_out, n_in, K, 0.4)
        x = rng.standard_normal((n_in, cin)).astype(np.float32)
        dy = rng.standard_normal((n_out, cout)).astype(np.float32)
        xb, gb = bf16_round(x), bf16_round(dy)
        want = np.zeros((K, cin, cout), np.float64)
        for k in range(K):
            rows = np.flatnonzero(nbr[:, k] >= 0)
            want[k] = xb[nbr[rows, k]].astype(np.float64).T @ gb[rows].astype(np.float64)
        xh, gh = bf16_bits(x), bf16_bits(dy)
        nf = int(emu.fns['es_spconv_wgrad_workspace_floats'](1, P(xh), 1, cin, P(gh), 1, cout, n_out, n_in, K, cin, cout))
        ws = np.zeros(max(nf, 1), np.float32)
        dw = np.zeros((K, cin, cout), np.float32)
        emu.launches()
        emu('es_spconv_wgrad_bf16_src', P(xh), 1, cin, P(gh), 1, cout, P(nbr), n_out, n_in, K, cin, cout, P(dw), 0, P(ws) if nf else 0,
            nf, 0)
        ran = emu.launches()
        kind = 'k_spconv_wgrad_bf16_tr' if tr else ('k_spconv_wgrad_bf16_big' if cin == 128 else 'k_spconv_wgrad_bf16<')
        assert any(kind in k for k in ran), (kind, ran)
        err = np.abs(dw - want).max() / np.abs(want).max()
        assert err < 3e-6, (cin, cout, tr, err)
    emu.lib.es_emu_set_dma_mode(0)
    emu('es_set_option', 14, 1)


@pytest.mark.parametrize('n', [300, 5000])
def test_norm_forward_backward(emu, n):
    """batch norm (train mode) + ReLU: n = 300 takes the one-launch column-block kernels, n = 5000 the chunked statistics ->
    finalize -> apply pipeline; forward values, saved statistics, bf16 shadow, and the backward pass (dx, dweight, dbias)"""
    rng = np.random.default_rng(n)
    C = 64
    x = (rng.standard_normal((n, C)) * 2 + 3).astype(np.float32)
    w, b = (rng.random(C) + 0.5).astype(np.float32), rng.standard_normal(C).astype(np.float32)
    seg = np.array([0, n], np.int32)
    mean, invstd = np.zeros(C, np.float32), np.zeros(C, np.float32)
    nws = int(emu.fns['es_norm_workspace_floats'](n, C, P(seg), 1))
    ws = np.zeros(max(nws, 1), np.float32)
    y = np.full((n, C), np.nan, np.float32)
    yh = np.zeros((n, C), np.uint16)
    rm, rv = np.zeros(C, np.float32), np.ones(C, np.float32)
    emu.launches()
    emu('es_norm_fwd', P(x), C, n, C, P(seg), 1, 1e-5, P(w), P(b), 0, 0, 1, P(rm), P(rv), 0.1, P(mean), P(invstd), P(ws), P(y), C,
        P(yh), 0)
    assert emu.launches() == (['k_norm_fwd_cb'] if n <= 4096 else ['k_norm_stats', 'k_norm_finalize', 'k_norm_apply4'])
    x64 = x.astype(np.float64)
    m, v = x64.mean(0), x64.var(0)
    want = np.maximum((x64 - m) / np.sqrt(v + 1e-5) * w + b, 0)
    assert np.abs(mean - m).max() < 1e-5 and np.abs(invstd * np.sqrt(v + 1e-5) - 1).max() < 1e-5
    assert np.abs(y - want).max() < 2e-5
    assert np.array_equal(yh, bf16_bits(y))
    assert np.abs(rm - 0.1 * m).max() < 1e-5 and np.abs(rv - (0.9 + 0.1 * v * n / (n - 1))).max() < 1e-4
    # backward: dy through ReLU, then the batch-norm backward of train mode
    dy = rng.standard_normal((n, C)).astype(np.float32)
    dz = np.where(want > 0, dy.astype(np.float64), 0.0)
    xhat = (x64 - m) / np.sqrt(v + 1e-5)
    dwt, dbt = (dz * xhat).sum(0), dz.sum(0)
    dxt = (dz - dbt / n - xhat * dwt / n) * w / np.sqrt(v + 1e-5)
    proto = emu.fns['es_norm_bwd'].argtypes
    assert len(proto) >= 10
    dyc = dy.copy()
    dx, dw, db = np.full((n, C), np.nan, np.float32), np.zeros(C, np.float32), np.zeros(C, np.float32)
    _norm_bwd(emu, dyc, y, x, n, C, seg, w, mean, invstd, ws, dw, db, dx)
    assert np.abs(dw - dwt).max() / np.abs(dwt).max() < 2e-5 and np.abs(db - dbt).max() / np.abs(dbt).max() < 2e-5
    assert np.abs(dx - dxt).max() / np.abs(dxt).max() < 2e-5


def _norm_bwd(emu, dy, y, x, n, C, seg, w, mean, invstd, ws, dw, db, dx):
    """es_norm_bwd by argument NAME (the header is the contract; this keeps the test readable if it grows an argument)"""
    from embodiedscan_amd import hip
    names = hip.PROTOS['es_norm_bwd'][2]
    # (backward workspace = the forward one + 2 * nseg * C floats for the reduced sums, as engine.batch_norm sizes it)
    wsb = np.zeros(int(emu.fns['es_norm_workspace_floats'](n, C, P(seg), 1)) + 2 * C, np.float32)
    val = dict(dy=P(dy), ldd=C, y=P(y), ldy=C, x=P(x), ldx=C, n=n, C=C, seg_off_host=P(seg), seg_off=P(seg), nseg=1, weight=P(w),
               mean=P(mean), invstd=P(invstd), act=1, dweight=P(dw), dbias=P(db), accumulate=0, workspace=P(wsb), dx=P(dx), ldo=C,
               dx_bf16=0, stream=0)
    missing = [a for a in names if a not in val]
    assert not missing, f'es_norm_bwd arguments this test does not know: {missing}'
    emu('es_norm_bwd', *[val[a] for a in names])


def test_radix_sort_and_topk_and_column_sums(emu):
    """the hand-written LSD radix sort (stable, int payload), the per-segment top-k mask -- single-workgroup kernel and the
    multi-workgroup histogram / election variant on one shared workspace with changing segment counts -- and the deterministic
    column sums with their last-workgroup election"""
    rng = np.random.default_rng(21)
    for n in (1, 77, 5000, 30000):
        keys = rng.integers(0, 1 << 40, n).astype(np.int64)
        keys[rng.integers(0, n, n // 3)] = keys[0]                      # duplicates: stability matters
        src = np.arange(n, dtype=np.int32)[::-1].copy()
        nb = int(emu.fns['es_sort_scratch_bytes'](n))
        scratch = np.zeros(nb + 64, np.uint8)
        ok, os_ = np.zeros(n, np.int64), np.zeros(n, np.int32)
        emu('es_sort_u64', P(keys), P(src), n, P(scratch), nb, P(ok), P(os_), 0)
        order = np.argsort(keys, kind='stable')
        assert np.array_equal(ok, keys[order]) and np.array_equal(os_, src[order]), n
    nw = int(emu.fns['es_topk_mask_workspace_ints'](1))
    ws = np.zeros(nw, np.int32)                                        # ONE workspace for every call below (fixed layout)
    for seg_sizes, k in (((3000,), 1000), ((50, 0, 1200, 999), 300), ((20000,), 5000), ((5,), 10)):
        off = np.concatenate([[0], np.cumsum(seg_sizes)]).astype(np.int32)
        n = int(off[-1])
        v = rng.standard_normal(n).astype(np.float32)
        v[rng.integers(0, n, n // 4)] = v[0]                            # ties: the lower index wins
        v[rng.integers(0, n, 5)] = -0.0
        want = np.zeros(n, np.int32)
        for s in range(len(seg_sizes)):
            a, b = off[s], off[s + 1]
            idx = np.lexsort((np.arange(b - a), -v[a:b].astype(np.float64)))[:k]
            want[a + idx] = 1
        m1, m2 = np.full(n, -1, np.int32), np.full(n, -1, np.int32)
        emu('es_topk_mask', P(v), P(off), len(seg_sizes), k, P(m1), 0)
        emu('es_topk_mask_ws', P(v), P(off), len(seg_sizes), k, P(m2), P(ws), nw, 0)
        assert np.array_equal(m1, want), (seg_sizes, k)
        assert np.array_equal(m2, want), (seg_sizes, k)
    for n, C in ((1, 8), (700, 64), (20000, 24), (100000, 4), (3072, 256), (300, 320), (130, 2048)):      # (1 .. 32 column groups over <= 4 workgroup columns)
        g = rng.standard_normal((n, C + 4)).astype(np.float32)[:, :C]   # strided rows
        ld = g.strides[0] // 4
        nf = int(emu.fns['es_colsum_workspace_floats'](n, C))
        w = np.zeros(nf, np.float32)
        dst = np.full(C, 7.0, np.float32)
        emu('es_colsum', P(g), ld, n, C, P(dst), 1, P(w), nf, 0)
        emu('es_colsum', P(g), ld, n, C, P(dst), 1, P(w), nf, 0)        # same workspace again: the ticket was left at zero
        want = 7.0 + 2 * g.astype(np.float64).sum(0)
        assert np.abs(dst - want).max() <= 2e-6 * np.abs(g).sum(0).max(), (n, C)
        assert not w[:4].view(np.int32).any()


def test_layernorm_and_attention(emu):
    rng = np.random.default_rng(31)
    n, C = 700, 256
    x, res = rng.standard_normal((n, C)).astype(np.float32), rng.standard_normal((n, C)).astype(np.float32)
    w, b = (rng.random(C) + 0.5).astype(np.float32), rng.standard_normal(C).astype(np.float32)
    y, z = np.zeros((n, C), np.float32), np.zeros((n, C), np.float32)
    mean, rstd = np.zeros(n, np.float32), np.zeros(n, np.float32)
    emu('es_layernorm_fwd', P(x), P(res), n, C, P(w), P(b), 1e-5, P(y), P(z), P(mean), P(rstd), 0)
    z64 = x.astype(np.float64) + res
    m, v = z64.mean(1, keepdims=True), z64.var(1, keepdims=True)
    xh = (z64 - m) / np.sqrt(v + 1e-5)
    assert np.abs(y - (xh * w + b)).max() < 1e-5 and np.array_equal(z, (x + res))
    dy = rng.standard_normal((n, C)).astype(np.float32)
    nf = int(emu.fns['es_layernorm_bwd_workspace_floats'](n, C))
    ws = np.zeros(nf, np.float32)
    dz, dw, db = np.zeros((n, C), np.float32), np.zeros(C, np.float32), np.zeros(C, np.float32)
    for _ in range(2):                                                  # dw / db accumulate (+=); the workspace is reused
        emu('es_layernorm_bwd', P(dy), P(z), n, C, P(w), P(mean), P(rstd), P(dz), 0, P(dw), P(db), P(ws), nf, 0)
    g = dy.astype(np.float64) * w
    dzt = (g - g.mean(1, keepdims=True) - xh * (g * xh).mean(1, keepdims=True)) / np.sqrt(v + 1e-5)
    assert np.abs(dz - dzt).max() < 2e-5
    assert np.abs(dw - 2 * (dy * xh).sum(0)).max() < 1e-3 and np.abs(db - 2 * dy.astype(np.float64).sum(0)).max() < 1e-3
    assert not ws[:4].view(np.int32).any()
    # attention core: head_dim 32, prefix key masks, both matrix-core types
    B, H, Lq, Lk = 2, 4, 70, 150
    D = 32 * H
    Q, K, V = (rng.standard_normal((B * L, D)).astype(np.float32) for L in (Lq, Lk, Lk))
    klen = np.array([Lk, 37], np.int32)
    want = np.zeros((B * Lq, D))
    for bb in range(B):
        for h in range(H):
            q = Q[bb * Lq:(bb + 1) * Lq, 32 * h:32 * h + 32].astype(np.float64)
            k = K[bb * Lk:bb * Lk + klen[bb], 32 * h:32 * h + 32].astype(np.float64)
            vv = V[bb * Lk:bb * Lk + klen[bb], 32 * h:32 * h + 32].astype(np.float64)
            s = q @ k.T / np.sqrt(32.0)
            p = np.exp(s - s.max(1, keepdims=True))
            want[bb * Lq:(bb + 1) * Lq, 32 * h:32 * h + 32] = (p / p.sum(1, keepdims=True)) @ vv
    for bf16, tol in ((0, 2e-5), (1, 2e-2)):
        O, lse = np.zeros((B * Lq, D), np.float32), np.zeros((B, H, Lq), np.float32)
        emu('es_attn_fwd', P(Q), D, P(K), D, P(V), D, B, H, Lq, Lk, P(klen), P(O), D, P(lse), bf16, 0)
        assert np.abs(O - want).max() < tol * max(1.0, np.abs(want).max()), (bf16, np.abs(O - want).max())


def test_device_point_sample_keys_follow_the_restated_generator(emu):
    """es_draw_keys (N4, counter-based PointSample draws) against oracle/draws.py: key per pixel, -1 for zero depth, the 64-bit
    sort key layout view << 54 | (2^30 - 1 - key) << 24 | pixel"""
    from oracle import draws as OD
    rng = np.random.default_rng(41)
    V, HW, seed = 3, 5000, 123456789
    depth = rng.random((V, HW)).astype(np.float32)
    depth[rng.random((V, HW)) < 0.2] = 0
    values, keys = np.zeros((V, HW), np.float32), np.zeros((V, HW), np.int64)
    emu('es_draw_keys', P(depth), V, HW, seed, P(values), P(keys), 0)
    pix = np.arange(HW, dtype=np.int64)
    for v in range(V):
        k30 = OD.key30(seed, v, pix)
        ok = depth[v] != 0
        assert np.all(values[v][~ok] == -1)
        assert np.array_equal(keys[v][ok], (np.int64(v) << 54) | ((np.int64((1 << 30) - 1) - k30[ok]) << 24) | pix[ok])
        assert np.array_equal(np.argsort(-values[v][ok].astype(np.float64), kind='stable'), np.argsort(-k30[ok], kind='stable'))


def test_convolution_entry_points_on_ragged_and_degenerate_shapes(emu):
    """random ODD shapes through the five convolution entry points (exact-f32 forward, bf16 forward from f32 and from bf16 rows,
    f32 and bf16 weight gradients): 1 .. 257 rows, 3 .. 128 input and 1 .. 192 output channels (no multiple of any tile size
    required), 1 .. 27 taps, maps from empty to full -- against f64 on the (rounded) operands; plus the empty launch.
    (220 such cases were run once with random thread schedules and DMA modes: no mismatch.)"""
    rng = np.random.default_rng(2024)
    y = np.full((4, 8), 5.0, np.float32)
    emu('es_spconv_fwd', 0, 8, 0, 0, 0, 0, 1, 8, 8, 0, P(y), 8, 0, 0, 0)            # n_out = 0: nothing happens
    assert (y == 5).all()
    for _ in range(14):
        n_out = int(rng.choice([1, 2, 15, 17, 63, 65, 129, 200, 257]))
        n_in = int(rng.choice([1, 3, 64, 130, 300]))
        K = int(rng.choice([1, 2, 8, 9, 27]))
        cin = int(rng.choice([3, 4, 8, 24, 32, 40, 64, 96, 128]))
        cout = int(rng.choice([1, 3, 8, 24, 32, 64, 72, 96, 128, 192]))
        nbr = _map(rng, n_out, n_in, K, float(rng.choice([0.0, 0.1, 0.5, 1.0])))
        x = rng.standard_normal((n_in, cin)).astype(np.float32)
        w = (rng.standard_normal((K, cin, cout)) / np.sqrt(K * cin)).astype(np.float32)
        bias = rng.standard_normal(cout).astype(np.float32) if rng.random() < 0.7 else None
        case = (n_out, n_in, K, cin, cout)
        want = _conv_ref(x, w, nbr, bias)
        y = np.full((n_out, cout), np.nan, np.float32)
        emu('es_spconv_fwd', P(x), cin, P(w), P(nbr), n_out, n_in, K, cin, cout, P(bias), P(y), cout, 0, 0, 0)
        assert np.abs(y - want).max() <= 1e-5 * max(np.abs(want).max(), 1e-3), case
        wt, wn = np.zeros((K, cout, cin), np.uint16), np.zeros((K, cin, cout), np.uint16)
        emu('es_cast_weight_bf16', P(w), K, cin, cout, P(wn), P(wt), 0)
        wantb = _conv_ref(bf16_round(x), bf16_round(w), nbr, bias)
        tol = 3e-6 * max(np.abs(wantb).max(), 1e-3)
        y = np.full((n_out, cout), np.nan, np.float32)
        emu('es_spconv_fwd_bf16', P(x), 0, cin, P(wt), P(nbr), n_out, n_in, K, cin, cout, P(bias), P(y), cout, 0, 0)
        assert np.abs(y - wantb).max() <= tol, case
        if emu.fns['es_spconv_bf16_is_fast'](n_in, cin, K, cin, cout):
            xh = bf16_bits(x)
            y = np.full((n_out, cout), np.nan, np.float32)
            emu('es_spconv_fwd_bf16', P(xh), 1, cin, P(wt), P(nbr), n_out, n_in, K, cin, cout, P(bias), P(y), cout, 0, 0)
            assert np.abs(y - wantb).max() <= tol, case
        dy = rng.standard_normal((n_out, cout)).astype(np.float32)
        for fn, xs, gs in (('es_spconv_wgrad', x, dy), ('es_spconv_wgrad_bf16', bf16_round(x), bf16_round(dy))):
            wantw = np.zeros((K, cin, cout))
            for k in range(K):
                rows = np.flatnonzero(nbr[:, k] >= 0)
                wantw[k] = xs[nbr[rows, k]].astype(np.float64).T @ gs[rows].astype(np.float64)
            dw = np.full((K, cin, cout), np.nan, np.float32)
            emu(fn, P(x), cin, P(dy), cout, P(nbr), n_out, n_in, K, cin, cout, P(dw), 0, 0, 0, 0)
            assert np.abs(dw - wantw).max() <= 1e-5 * max(np.abs(wantw).max(), 1e-3), (fn, case)


def test_narrow_input_convolution_kernels(emu):
    """MinkResNet.conv1's shape (K = 27, 3 -> 64 channels, exact f32): the lane-per-output-channel forward / weight-gradient kernels
    (option 21, default on) against f64 and against the tiled kernels they replace -- ragged last tile, rows with no neighbour, a
    tap nobody uses, bias, accumulate, the row-sliced workspace form and the single-slice form, strided dY rows"""
    rng = np.random.default_rng(31)
    for n_out, n_in, fill in ((1, 5, 1.0), (130, 77, 0.2), (64, 300, 0.0), (517, 400, 0.35)):
        K, cin, cout = 27, 3, 64
        nbr = _map(rng, n_out, n_in, K, fill)
        nbr[:, 11] = -1
        if n_out > 3:
            nbr[3] = -1
        x = rng.standard_normal((n_in, cin)).astype(np.float32)
        w = (rng.standard_normal((K, cin, cout)) / 9).astype(np.float32)
        bias = rng.standard_normal(cout).astype(np.float32)
        want = _conv_ref(x, w, nbr, bias)
        tol = 1e-5 * max(np.abs(want).max(), 1e-3)
        got = {}
        for on in (1, 0):
            emu('es_set_option', 21, on)
            emu.launches()
            y = np.full((n_out, cout + 8), np.nan, np.float32)
            emu('es_spconv_fwd', P(x), cin, P(w), P(nbr), n_out, n_in, K, cin, cout, P(bias), P(y), cout + 8, 0, 0, 0)
            assert any('k_spconv_narrow_fwd' in k for k in emu.launches()) == bool(on)
            assert np.abs(y[:, :cout] - want).max() <= tol and np.isnan(y[:, cout:]).all(), (n_out, on)
            got[on] = y[:, :cout].copy()
            y2 = np.ascontiguousarray(y[:, :cout]) * 0 + 1
            emu('es_spconv_fwd', P(x), cin, P(w), P(nbr), n_out, n_in, K, cin, cout, 0, P(y2), cout, 0, 1, 0)       # accumulate, no bias
            assert np.abs(y2 - 1 - (want - bias)).max() <= tol
        assert np.abs(got[1] - got[0]).max() <= tol
        dyb = rng.standard_normal((n_out, cout + 4)).astype(np.float32)
        dy = dyb[:, :cout]
        wantw = np.zeros((K, cin, cout))
        for k in range(K):
            rows = np.flatnonzero(nbr[:, k] >= 0)
            wantw[k] = x[nbr[rows, k]].astype(np.float64).T @ dy[rows].astype(np.float64)
        tolw = 1e-5 * max(np.abs(wantw).max(), 1e-3)
        for on in (1, 0):
            emu('es_set_option', 21, on)
            nf = int(emu.fns['es_spconv_wgrad_workspace_floats'](0, 0, 0, cin, 0, 0, cout + 4, n_out, n_in, K, cin, cout))
            ws = np.full(max(nf, 1), np.nan, np.float32)
            for use_ws in (1, 0):
                emu.launches()
                dw = np.full((K, cin, cout), 2.0, np.float32)
                emu('es_spconv_wgrad', P(x), cin, P(dy), cout + 4, P(nbr), n_out, n_in, K, cin, cout, P(dw), 1, P(ws) if use_ws else 0, nf if use_ws else 0, 0)
                assert any('k_spconv_narrow_wgrad' in k for k in emu.launches()) == bool(on)
                assert np.abs(dw - 2 - wantw).max() <= tolw, (n_out, on, use_ws)
    emu('es_set_option', 21, 1)


def test_segmented_norm_and_attention_on_odd_shapes(emu):
    """instance-norm style segments (1 .. 12 per call, sizes 1 .. 4097 incl. EMPTY segments), channel counts that are no multiple
    of 4 (scalar paths), strided rows, residual, ReLU / ELU, forward and backward; attention with odd query / key counts and
    per-sample key lengths down to 1, on f32 and bf16 matrix cores.  (110 random cases of this kind were run once: no
    mismatch beyond f32 cancellation in 1- and 2-row segments.)"""
    rng = np.random.default_rng(77)

    def act_f(z, act):
        return np.maximum(z, 0) if act == 1 else (np.where(z > 0, z, np.expm1(np.minimum(z, 0))) if act == 2 else z)
    for sizes, C, pad, act, use_res in (((7,), 3, 0, 0, False), ((4097, 0, 129), 24, 0, 1, True), ((500, 64, 129, 7, 64, 33, 500, 9, 64, 129, 500, 70), 64, 8, 2, True),
                                        ((129, 4097), 100, 4, 1, False)):
        nseg = len(sizes)
        off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
        n, ldx = int(off[-1]), C + pad
        xb = (rng.standard_normal((n, ldx)) * 2 + 1).astype(np.float32)
        x = xb[:, :C]
        w, b = (rng.random(C) + .5).astype(np.float32), rng.standard_normal(C).astype(np.float32)
        res = rng.standard_normal((n, C)).astype(np.float32)
        mean, invstd = np.zeros((nseg, C), np.float32), np.zeros((nseg, C), np.float32)
        nws = int(emu.fns['es_norm_workspace_floats'](n, C, P(off), nseg)) + 2 * nseg * C
        ws, ws2 = np.zeros(nws, np.float32), np.zeros(nws, np.float32)
        y = np.full((n, C), np.nan, np.float32)
        emu('es_norm_fwd', P(xb), ldx, n, C, P(off), nseg, 1e-5, P(w), P(b), P(res) if use_res else 0, C, act, 0, 0, 0.1, P(mean),
            P(invstd), P(ws), P(y), C, 0, 0)
        want, xhat, istd = np.zeros((n, C)), np.zeros((n, C)), np.zeros((nseg, C))
        for s in range(nseg):
            a, e = off[s], off[s + 1]
            if e > a:
                xs = x[a:e].astype(np.float64)
                istd[s] = 1 / np.sqrt(xs.var(0) + 1e-5)
                xhat[a:e] = (xs - xs.mean(0)) * istd[s]
        z = xhat * w + b + (res if use_res else 0)
        want = act_f(z, act)
        assert np.abs(y - want).max() < 5e-5 * max(1, np.abs(want).max()), (sizes, C)
        dy = rng.standard_normal((n, C)).astype(np.float32)
        dz = dy.astype(np.float64) * (1.0 if act == 0 else ((z > 0) if act == 1 else np.where(z > 0, 1.0, np.exp(np.minimum(z, 0)))))
        dxt = np.zeros((n, C))
        for s in range(nseg):
            a, e = off[s], off[s + 1]
            if e > a:
                ds, xs = dz[a:e], xhat[a:e]
                dxt[a:e] = (ds - ds.mean(0) - xs * (ds * xs).mean(0)) * w * istd[s]
        dx, dw, db = np.full((n, C), np.nan, np.float32), np.zeros(C, np.float32), np.zeros(C, np.float32)
        dyc = dy.copy()
        emu('es_norm_bwd', P(dyc), C, P(y), C, P(xb), ldx, n, C, P(off), nseg, P(mean), P(invstd), P(w), act, P(dw), P(db), P(ws2), P(dx), C,
            0, 0, 0)
        assert np.abs(dx - dxt).max() < 3e-4 * np.abs(dxt).max() and np.abs(dw - (dz * xhat).sum(0)).max() < 3e-4 * np.abs((dz * xhat).sum(0)).max()
        assert np.abs(db - dz.sum(0)).max() < 3e-4 * np.abs(dz.sum(0)).max()
    for B, H, Lq, Lk, klen in ((1, 1, 1, 1, None), (3, 2, 17, 33, (33, 1, 7)), (2, 8, 100, 130, (64, 130)), (1, 2, 256, 300, None)):
        D = 32 * H
        Q, K, V = (rng.standard_normal((B * L, D)).astype(np.float32) for L in (Lq, Lk, Lk))
        kl = np.array(klen if klen else [Lk] * B, np.int32)
        want = np.zeros((B * Lq, D))
        for bb in range(B):
            for h in range(H):
                q = Q[bb * Lq:(bb + 1) * Lq, 32 * h:32 * h + 32].astype(np.float64)
                k = K[bb * Lk:bb * Lk + kl[bb], 32 * h:32 * h + 32].astype(np.float64)
                v = V[bb * Lk:bb * Lk + kl[bb], 32 * h:32 * h + 32].astype(np.float64)
                s = q @ k.T / np.sqrt(32.0)
                p = np.exp(s - s.max(1, keepdims=True))
                want[bb * Lq:(bb + 1) * Lq, 32 * h:32 * h + 32] = (p / p.sum(1, keepdims=True)) @ v
        for bf16, tol in ((0, 3e-5), (1, 3e-2)):
            O, lse = np.full((B * Lq, D), np.nan, np.float32), np.zeros((B, H, Lq), np.float32)
            emu('es_attn_fwd', P(Q), D, P(K), D, P(V), D, B, H, Lq, Lk, P(kl) if klen else 0, P(O), D, P(lse), bf16, 0)
            assert np.abs(O - want).max() < tol * max(1, np.abs(want).max()), (B, H, Lq, Lk, bf16)
