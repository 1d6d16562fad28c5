"""The next round's 256-row gather-convolution tile (embodiedscan_amd/csrc/next/spconv_tile256.hip: NOT in libes_hip.so, never
run on a GPU) under the CDNA emulator: bit-identical to the shipped LDS-DMA kernel on the same operands -- f32 and bf16 output
rows, every epilogue mode, ragged tiles, unused taps, the identity map -- under two thread schedules and with late LDS-DMA
delivery, and within the gfx950 register / LDS budget when compiled by hipcc."""
import ctypes
import os
import re
import shutil
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests', 'emu'))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
CSRC = os.path.join(ROOT, 'embodiedscan_amd', 'csrc')
HIPCC = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'


@pytest.fixture(scope='module')
def lib():
    import build as emu_build
    from embodiedscan_amd import hip
    lib = ctypes.CDLL(emu_build.build(files=('spconv.hip', 'rowops.hip', 'next/spconv_tile256.hip'), lib_name='libes_emu_next.so'))
    for name, (ret, at, _) in hip.PROTOS.items():
        f = getattr(lib, name, None)
        if f is not None:
            f.restype, f.argtypes = ret, at
    V, I, F = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
    lib.es_next_spconv_fwd_bf16_tile.restype = I
    lib.es_next_spconv_fwd_bf16_tile.argtypes = [V, I, V, V, I, I, I, I, I, V, V, I, I, V, V, V, I, I, I, I, I, I, V]
    lib.es_emu_set_schedule.argtypes = [ctypes.c_int, ctypes.c_ulonglong]
    return lib


@pytest.mark.parametrize('schedule,lazy', [(1, 0), (2, 1)])
def test_256_row_tile_is_bit_identical_to_the_shipped_dma_kernel(lib, schedule, lazy):
    from test_emu_kernels import P, _conv_ref, _map, bf16_bits, bf16_round
    lib.es_emu_set_schedule(schedule, 99)
    lib.es_emu_set_dma_mode(lazy)
    rng = np.random.default_rng(3)
    n_checked = 0
    try:
        shapes = ((380, 350, 27, 64, 128, 0.35, False), (300, 300, 27, 128, 64, 0.5, False), (515, 515, 1, 128, 128, 1.0, True),
                  (260, 300, 8, 64, 256, 0.9, False))
        for n_out, n_in, K, cin, cout, fill, ident in (shapes if not lazy else shapes[:2]):
            nbr = None if ident else _map(rng, n_out, n_in, K, fill)
            if nbr is not None and K > 5:
                nbr[:, 5] = -1
            x = rng.standard_normal((n_in, cin)).astype(np.float32)
            w = (rng.standard_normal((K, cin, cout)) / np.sqrt(K * cin)).astype(np.float32)
            xh = bf16_bits(x)
            wt, wn = np.zeros((K, cout, cin), np.uint16), np.zeros((K, cin, cout), np.uint16)
            assert lib.es_cast_weight_bf16(P(w), K, cin, cout, P(wn), P(wt), 0) == 0
            bias = rng.standard_normal(cout).astype(np.float32)
            scale, shift = (rng.random(cout) + 0.5).astype(np.float32), rng.standard_normal(cout).astype(np.float32)
            res = rng.standard_normal((n_out, cout)).astype(np.float32)
            resh = bf16_bits(res)
            want = _conv_ref(bf16_round(x), bf16_round(w), nbr if nbr is not None else np.arange(n_out, dtype=np.int32)[:, None], bias)
            for chunk, cols in ((2, 0), (1, 0)) + (((2, 256),) if cout % 256 == 0 else ()):
                if cout % 128 and chunk == 1:
                    continue                                  # (64 columns x 32 channels: fewer pieces than threads)
                lib.es_set_option(10, chunk)
                lib.es_set_option(11, 0)
                lib.es_set_option(3, 0)                       # K = 1: keep the shipped path on the conv kernel, not the row GEMM
                # plain: bias, f32 rows
                y0 = np.full((n_out, cout), np.nan, np.float32)
                assert lib.es_spconv_fwd_bf16(P(xh), 1, cin, P(wt), P(nbr), n_out, n_in, K, cin, cout, P(bias), P(y0), cout, 0, 0) == 0
                y1 = np.full((n_out, cout), np.nan, np.float32)
                rc = lib.es_next_spconv_fwd_bf16_tile(P(xh), cin, P(wt), P(nbr), n_out, n_in, K, cin, cout, P(bias), P(y1), cout, 0,
                                                      0, 0, 0, 0, 0, 0, 256, cols, chunk, 0)
                assert rc == 0
                assert np.array_equal(y0, y1), (n_out, cin, cout, chunk, float(np.abs(y0 - y1).max()))
                assert np.abs(y1 - want).max() / np.abs(want).max() < 2e-6
                n_checked += 1
                # accumulate into Y
                y2, y3 = res.copy(), res.copy()
                assert lib.es_spconv_fwd_bf16(P(xh), 1, cin, P(wt), P(nbr), n_out, n_in, K, cin, cout, 0, P(y2), cout, 1, 0) == 0
                assert lib.es_next_spconv_fwd_bf16_tile(P(xh), cin, P(wt), P(nbr), n_out, n_in, K, cin, cout, 0, P(y3), cout, 1,
                                                        0, 0, 0, 0, 0, 0, 256, cols, chunk, 0) == 0
                assert np.array_equal(y2, y3)
                # fused epilogues of the image backbone: (act, residual kind, bf16 residual?, bf16 output?)
                modes = ((1, res, 0, 0), (0, None, 0, 0), (3, res, 0, 0), (1, resh, 1, 1), (1, None, 0, 1), (3, resh, 1, 0))
                for act, r, rh, yh in (modes if not lazy and chunk == 2 else modes[3:4]):
                    dt = np.uint16 if yh else np.float32
                    ya, yb = np.zeros((n_out, cout), dt), np.zeros((n_out, cout), dt)
                    assert lib.es_spconv_fwd_bf16_io(P(xh), 1, cin, P(wt), P(nbr), n_out, n_in, K, cin, cout, P(scale),
                                                     P(shift) if act != 3 else 0, P(r), rh, cout if r is not None else 0, act, P(ya), yh,
                                                     cout, 0) == 0
                    assert lib.es_next_spconv_fwd_bf16_tile(P(xh), cin, P(wt), P(nbr), n_out, n_in, K, cin, cout, 0, P(yb), cout, 0,
                                                            P(scale), P(shift) if act != 3 else 0, P(r), cout if r is not None else 0,
                                                            act, (1 if yh else 0) | (2 if rh else 0), 256, cols, chunk, 0) == 0
                    assert np.array_equal(ya, yb), (act, rh, yh, chunk)
                    n_checked += 1
    finally:
        lib.es_emu_set_schedule(0, 1)
        lib.es_emu_set_dma_mode(0)
        lib.es_set_option(10, 2)
        lib.es_set_option(11, 768)
        lib.es_set_option(3, 1)
    print(f'256-row tile: {n_checked} outputs bit-identical to the shipped LDS-DMA kernel')


@pytest.mark.skipif(not os.path.exists(HIPCC), reason='hipcc not installed')
def test_256_row_tile_fits_gfx950():
    """hipcc for gfx950: no spills / scratch; 64-channel chunks: 123.6 KB of LDS (one 8-wave workgroup per CU), 32-channel
    chunks: 75.6 KB (two)"""
    out = subprocess.run([HIPCC, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=off', '-Wno-unused-result',
                          '-Rpass-analysis=kernel-resource-usage', '-I', CSRC, '-c', os.path.join(CSRC, 'next', 'spconv_tile256.hip'),
                          '-o', os.devnull], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    kernels, cur = {}, None
    for line in out.stderr.splitlines():
        m = re.search(r'remark: Function Name: (\S+)', line)
        if m:
            cur = kernels.setdefault(m.group(1), {})
            continue
        m = re.search(r'remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+) \[-Rpass', line)
        if m and cur is not None:
            cur[m.group(1).strip()] = int(m.group(2))
    assert len(kernels) == 9
    for name, r in kernels.items():
        print(name[:44], {k: r[k] for k in ('VGPRs', 'AGPRs', 'ScratchSize', 'Occupancy', 'LDS Size') if k in r})
        assert r.get('ScratchSize', 0) == 0 and r.get('VGPRs Spill', 0) == 0 and r.get('SGPRs Spill', 0) == 0, (name, r)
        assert r.get('LDS Size', 0) <= 160 * 1024
    big = next(r for n, r in kernels.items() if 'ILi256ELi128ELi2E' in n)
    assert big['LDS Size'] <= 128 * 1024 and big['VGPRs'] <= 128
    wide = next(r for n, r in kernels.items() if 'ILi256ELi256ELi2E' in n)
    assert wide['LDS Size'] <= 160 * 1024 and wide['VGPRs'] + wide.get('AGPRs', 0) <= 256      # one 8-wave workgroup per CU: 2 waves per SIMD
    mid = next(r for n, r in kernels.items() if 'ILi256ELi128ELi1E' in n)
    assert mid['LDS Size'] <= 80 * 1024 and mid['VGPRs'] <= 128       # two workgroups = 16 waves per CU need <= 128 VGPRs


def test_staging_work_per_mfma_of_the_tiles(lib):
    """the work model behind the 256-row tile (DESIGN.md "Next round" 1), counted by the emulator on one 27-tap 128 -> 128 layer:
    LDS-DMA pieces per MFMA = 16 (M + N) / (M N) -- 0.25 for the shipped 128 x 128 tile, 0.1875 at 256 x 128 -- and half the
    barriers per MFMA"""
    from test_emu_kernels import P, _map, bf16_bits
    rng = np.random.default_rng(8)
    n, K, c = 512, 27, 128
    nbr = _map(rng, n, n, K, 1.0)                       # every neighbour present: no tile skips a tap
    xh = bf16_bits(rng.standard_normal((n, c)).astype(np.float32))
    w = rng.standard_normal((K, c, c)).astype(np.float32)
    wt, wn = np.zeros((K, c, c), np.uint16), np.zeros((K, c, c), np.uint16)
    lib.es_cast_weight_bf16(P(w), K, c, c, P(wn), P(wt), 0)
    y = np.zeros((n, c), np.float32)
    cnt = (ctypes.c_double * 3)()
    got = {}
    for rows in (128, 256):
        lib.es_emu_take_counters(cnt)
        assert lib.es_next_spconv_fwd_bf16_tile(P(xh), c, P(wt), P(nbr), n, n, K, c, c, 0, P(y), c, 0, 0, 0, 0, 0, 0, 0, rows, 0, 2, 0) == 0
        lib.es_emu_take_counters(cnt)
        got[rows] = (cnt[0], cnt[1], cnt[2])
    for rows, (mfma, dma, bar) in got.items():
        assert mfma == n // 16 * (c // 16) * K * (c // 32)                      # the same arithmetic either way
        assert abs(dma / mfma - 16 * (rows + 128) / (rows * 128)) < 1e-9, (rows, dma / mfma)
    assert got[256][2] / got[128][2] < 1.05                                     # barrier arrivals of waves: as many waves arrive half as often
    print({r: dict(mfma=v[0], dma_pieces=v[1], pieces_per_mfma=round(v[1] / v[0], 4), wave_barrier_arrivals=v[2]) for r, v in got.items()})
