"""The whole mv-3ddet loss forward -- image preprocessing, ResNet-50(w16), voxelisation, MinkResNet34, projection fusion, the
FCAF3D head with pruning, target assignment, the three losses -- with the product's own host code driving the kernel SOURCES
under the CDNA emulator of tests/emu, on CPU tensors, against the CPU oracle.  Development / audit tool (about ten minutes for a
2-view 60x80 scan in exact-f32 mode: the emulated f32 matrix-core tile is a wave rendezvous per 16x16x4 step); the CPU test suite
runs the kernel-level pieces (tests/test_emu_*.py).  Nothing here is a product path: the table swap below is what
tests/test_emu_product.py's fixture does, and hip.py itself binds libes_hip.so only.
    python tools/emu_forward.py        (result of the round-4 run: profiles/r4_emulated_forward.txt)"""
import ctypes
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'emu'))


def main():
    import torch
    import build as emu_build
    from embodiedscan_amd import hip, sparse
    lib = ctypes.CDLL(emu_build.build())
    fns = {}
    for name, (ret, at, _) in hip.PROTOS.items():
        f = getattr(lib, name)
        f.restype, f.argtypes = ret, at
        fns[name] = f
    stream = types.SimpleNamespace(cuda_stream=0, synchronize=lambda: None, wait_event=lambda e: None, wait_stream=lambda s: None)
    hip._fn, hip._STREAM, hip._STREAM_OBJ, hip.refresh_stream = fns, [0], [stream], (lambda: 0)
    sparse.read_ints = lambda t: [int(v) for v in t.reshape(-1).tolist()]
    torch.cuda.synchronize = lambda *a, **k: None
    torch.cuda.current_stream = lambda *a, **k: stream
    from embodiedscan_amd import engine as E, pipeline
    E.TWO_STREAMS[0] = E.WGRAD_ASYNC[0] = E.GRAPHS[0] = False          # the single-stream schedule
    from embodiedscan_amd.config import build_detector, load_config
    from embodiedscan_amd.synth import make_scan
    from oracle import model as OM
    dev = torch.device('cpu')
    det = build_detector(load_config(os.path.join(ROOT, 'configs', 'mv_3ddet.py')), device=dev, seed=0).to(dev)
    scan = make_scan(7, n_views=2, height=60, width=80, img_size=(64, 64), n_points=2500, n_boxes=5)
    batch = pipeline.make_batch([pipeline.upload_scan(scan, dev)])
    sd = det.state_dict()
    pts = [p.cpu() for p in batch['inputs']['points']]
    t0 = time.time()
    E.TAPE.clear()
    data = det.data_preprocessor(batch, True)
    det._bind()
    losses = det.forward(data['inputs'], data['data_samples'], mode='loss')
    dt = time.time() - t0
    imgs = OM.preprocess_img(torch.from_numpy(scan['img']), [123.675, 116.28, 103.53], [58.395, 57.12, 57.375])[None]
    ol = OM.detector_loss({k: v.cpu() for k, v in sd.items()}, pts, imgs, [scan['meta']], [torch.from_numpy(scan['gt_boxes'])],
                          [torch.from_numpy(scan['gt_labels'])])
    print(f'emulated forward: {dt:.0f} s')
    worst = 0.0
    for k in ol:
        a, b = float(losses[k]), float(ol[k])
        worst = max(worst, abs(a - b) / max(abs(b), 1e-6))
        print(f'{k}: emulated kernels {a:.8f}  oracle {b:.8f}')
    print(f'worst relative difference {worst:.2e}')
    assert worst < 1e-5


if __name__ == '__main__':
    main()
