"""The GPU suite's own kernel-level parity tests, run in the CPU suite: the product's HOST layer (embodiedscan_amd/sparse.py,
the head's target assignment, the fusion meta tables ...) drives the kernel sources under the CDNA emulator of tests/emu
instead of libes_hip.so, on CPU tensors (random thread schedule, late LDS-DMA delivery: the strictest modes).  The test BODIES are the ones of tests/test_gpu_*.py, imported and called unchanged --
what the MI355X must satisfy at round end is checked here first, bit-exact integer work included (voxelisation, kernel / stride
/ generative / union maps, target assignment against the reference's own output).
How: a test-only fixture swaps the ctypes table of embodiedscan_amd.hip for the emulated library's (same C ABI), pins the
stream handle to 0 and gives torch.cuda's synchronisation calls no-op stand-ins.  Nothing of this exists outside the fixture:
the product binds libes_hip.so and has no CPU path."""
import ctypes
import os
import sys
import types

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests', 'emu'))
sys.path.insert(0, os.path.join(ROOT, 'tests'))


class _ListAsDict:
    """monkeypatch.setitem for a one-element list (the engine's mutable flags)"""

    def __init__(self, lst):
        self.lst = lst

    def get(self, k, default=None):
        return self.lst[k]

    def __setitem__(self, k, v):
        self.lst[k] = v

    def __delitem__(self, k):
        pass


@pytest.fixture
def emulated(monkeypatch):
    import build as emu_build
    from embodiedscan_amd import hip, sparse
    lib = ctypes.CDLL(emu_build.build())
    fns = {}
    for name, (ret, at, _) in hip.PROTOS.items():
        f = getattr(lib, name)
        f.restype, f.argtypes = ret, at
        fns[name] = f
    # thread schedule between synchronisation points / LDS-DMA delivery (tests/emu): ES_EMU_SCHEDULE=0|1|2, ES_EMU_LAZY_DMA=0|1
    lib.es_emu_set_schedule.argtypes = [ctypes.c_int, ctypes.c_ulonglong]
    lib.es_emu_set_schedule(int(os.environ.get('ES_EMU_SCHEDULE', '2')), 777)
    lib.es_emu_set_dma_mode(int(os.environ.get('ES_EMU_LAZY_DMA', '1')))
    monkeypatch.setattr(hip, '_fn', fns)
    monkeypatch.setattr(hip, '_STREAM', [0])
    stream = types.SimpleNamespace(cuda_stream=0, synchronize=lambda: None, wait_event=lambda e: None, wait_stream=lambda s: None)
    monkeypatch.setattr(hip, '_STREAM_OBJ', [stream])
    monkeypatch.setattr(hip, 'refresh_stream', lambda: 0)
    monkeypatch.setattr(sparse, 'read_ints', lambda t: [int(v) for v in t.reshape(-1).tolist()])
    monkeypatch.setattr(torch.cuda, 'synchronize', lambda *a, **k: None)
    monkeypatch.setattr(torch.cuda, 'current_stream', lambda *a, **k: stream)
    # the engine's single-stream schedule (what ES_TWO_STREAMS=0 ES_WGRAD_ASYNC=0 ES_GRAPHS=0 select on the GPU)
    from embodiedscan_amd import engine as E
    for flag in (E.TWO_STREAMS, E.WGRAD_ASYNC, E.GRAPHS):
        monkeypatch.setitem(_ListAsDict(flag), 0, False)
    # the engine allocates the weight-gradient slice workspace of a stream on 'cuda' when it has none: hand it a host buffer
    E.drop_caches()                      # nothing cached by an earlier (GPU or emulated) test survives into this one ...
    monkeypatch.setitem(E._WGRAD_WS, 0, torch.empty(1 << 24, dtype=torch.float32))
    yield torch.device('cpu')
    E.drop_caches()                      # ... and no host buffer of this one is handed to a later GPU test of the same process


def test_voxelise_and_every_coordinate_map_bit_exact(emulated):
    import test_gpu_ops as T
    T.test_voxelize_and_maps_bit_exact(emulated)


def test_strided_chain(emulated):
    import test_gpu_ops as T
    T.test_strided_chain_one_round_trip_equals_the_level_by_level_chain(emulated)


def test_union_add_gather_topk(emulated):
    import test_gpu_ops as T
    T.test_union_add_and_gather(emulated)
    T.test_topk_mask(emulated)


def test_target_assignment_against_the_reference_output(emulated, golden_dir):
    import test_gpu_ops as T
    T.test_get_targets_golden_and_random(emulated, golden_dir)


# (tests/test_gpu_ops.py::test_rowgemm_matches_general_kernel also passes under emulation -- 7 epilogue modes x 10 shapes, 60 s; the
#  row GEMMs are covered in tests/test_emu_kernels.py, so it is not repeated here to keep the CPU suite short)


@pytest.mark.parametrize('bf16,tol', [(0, 2e-5), (1, 2e-2)])
def test_attention_forward_backward_vs_torch(emulated, bf16, tol):
    import test_gpu_grounding as T
    T.test_attention_fwd_bwd_vs_torch(emulated, bf16, tol)


def test_grounding_head_kernels_vs_torch_and_reference(emulated):
    """LayerNorm / ContrastiveEmbed / box decode / sorted top-k vs torch, the FCAF coder vs the reference's golden, rotated 3-D
    IoU vs the oracle, Hungarian matching identical to scipy on the reference's cost matrices + both losses and their gradients
    vs the reference's own outputs (the quad reductions inside divergent branches exercise the emulator's EXEC-mask model)"""
    import test_gpu_grounding as T
    T.test_layernorm_contrastive_decode_topk_vs_torch(emulated)
    T.test_fcaf_box_coder_vs_reference(emulated)
    T.test_box3d_iou_vs_oracle(emulated)
    T.test_matching_and_losses_vs_reference(emulated)


def test_occupancy_targets_and_losses_vs_reference(emulated):
    import test_gpu_occ as T
    T.test_occ_targets_and_losses_vs_reference(emulated)


def test_coordinate_path_on_adversarial_clouds(emulated):
    """voxelise -> strided chain -> kernel / inverse maps -> generative children -> union, bit-exact against oracle/coords.py on
    clouds the benchmark never produces: 12 samples of 1 .. 4000 points, coordinates on exact voxel boundaries (heavy duplicates),
    a whole sample inside ONE voxel, voxel indices spread over +-1000.  (76 random clouds of these kinds were run once under
    random thread schedules: no mismatch.)"""
    import numpy as np
    from embodiedscan_amd import sparse
    from oracle import coords as C
    rng = np.random.default_rng(12)
    for nb, vs, mode in ((12, 0.5, 1), (4, 0.04, 2), (2, 0.01, 3), (1, 0.04, 0)):
        pts = []
        for b in range(nb):
            n = int(rng.choice([1, 2, 50, 700, 2000]))
            if mode == 0:
                p = rng.random((n, 3)) * 4 - 2
            elif mode == 1:
                p = np.round(rng.random((n, 3)) * 6 - 3, 1)
            elif mode == 2:
                p = np.tile(rng.random((1, 3)), (n, 1))
            else:
                p = (rng.random((n, 3)) - 0.5) * 2 * 1000 * vs
            pts.append(p.astype(np.float32))
        oc, osrc = C.voxelize(pts, vs)
        cs, src = sparse.voxelize([torch.from_numpy(p) for p in pts], vs)
        assert cs.n == oc.shape[0] and np.array_equal(cs.coords.numpy(), oc) and np.array_equal(src.numpy().astype(np.int64), osrc)
        cur, ocur, ts = cs, oc, 1
        for stride, ks in ((2, 3), (2, 2), (2, 3)):
            out, oout = cur.strided(stride), C.stride_coords(ocur, ts * stride)
            assert np.array_equal(out.coords.numpy(), oout)
            onbr = C.kernel_map(ocur, oout, ks, ts)
            assert np.array_equal(cur.kernel_map(out, ks).numpy(), onbr)
            assert np.array_equal(cur.inverse_map(out, ks).numpy(), C.inverse_map(onbr, ocur.shape[0]))
            assert np.array_equal(out.kernel_map(out, 3).numpy(), C.kernel_map(oout, oout, 3, ts * stride))
            cur, ocur, ts = out, oout, ts * stride
        ch, och = cur.children(), C.gen_transpose_coords(ocur, ts)
        assert np.array_equal(ch.coords.numpy(), och)
        fine, ofine = cs.strided(2).strided(2), C.stride_coords(C.stride_coords(oc, 2), 4)
        u, pa, pb = sparse.union(fine, ch)
        ou, opa, opb = C.union_coords(ofine, och, nb)
        assert np.array_equal(u.coords.numpy(), ou) and np.array_equal(pa.numpy(), opa) and np.array_equal(pb.numpy(), opb)


def test_tape_level_operators_forward_and_backward(emulated, golden_dir):
    """engine-level (autograd tape) tests of the GPU suite in exact-f32 mode: generative transposed convolution + batch norm +
    max pooling forward AND backward against the oracle's autograd, the occupancy neck (dense 3-D convolutions, transposed
    up-convolution) against the reference module's golden output and gradients, the FPN against the oracle.
    (tests/test_gpu_ops.py::test_spconv_fwd_bwd -- convolution forward, data and weight gradients vs autograd -- runs below for its
    strided K = 1 shape; the 27-tap shapes also pass, at 1 - 2 min each under the random schedule.)"""
    import test_gpu_occ as TO
    import test_gpu_ops as T
    T.test_gen_transpose_norm_pool(emulated)
    TO.test_imvoxel_neck_vs_reference(emulated)
    TO.test_fpn_vs_oracle(emulated)


def test_convolution_forward_data_and_weight_gradients_vs_autograd(emulated):
    import test_gpu_ops as T
    T.test_spconv_fwd_bwd(emulated, 64, 128, 1, 2)
