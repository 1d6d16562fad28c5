"""Whole-model audits on the CPU: the product's own host code (engine, models, sparse layer) drives the kernel SOURCES under the
CDNA emulator of tests/emu, on CPU tensors, and the result is compared with the CPU oracle.

    python tools/emu_audit.py forward     the mv-3ddet loss forward: preprocessing, ResNet-50(w16), voxelisation, MinkResNet34,
                                          projection fusion, FCAF3D head with pruning, target assignment, losses  (~10 min)
    python tools/emu_audit.py train       tests/test_gpu_model.py::test_train_step_parity on a small batch: forward AND backward,
                                          integer outputs bit exact, every parameter gradient against the f64-calibrated oracle
    python tools/emu_audit.py predict     mode='predict' end to end (eval-mode norms, score top-k, decode, multi-class rotated NMS):
                                          detections identical to the oracle's
    python tools/emu_audit.py grounder    tests/test_gpu_grounding.py::test_grounder_train_step_vs_oracle[f32]: queries and
                                          Hungarian assignments identical, logits, 12 losses, 245 gradients

Development / audit tool: minutes to an hour per run in exact-f32 mode (the emulated f32 matrix-core tile is a wave rendezvous
per 16x16x4 step); the CPU test suite runs the kernel- and operator-level pieces (tests/test_emu_*.py).  Results of the round-4
runs: profiles/r4_emulated_*.txt.  Nothing here is a product path: emulate() is the table swap of tests/test_emu_product.py's
fixture (the ctypes table of embodiedscan_amd.hip -> the emulated library, stream handle 0, no-op stand-ins for torch.cuda's
synchronisation calls, a host buffer where the engine would allocate its weight-gradient workspace on 'cuda'); hip.py itself binds
libes_hip.so only and raises without it."""
import ctypes
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'emu'))
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def emulate():
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
    from embodiedscan_amd import engine as E
    E.TWO_STREAMS[0] = E.WGRAD_ASYNC[0] = E.GRAPHS[0] = False          # the single-stream schedule
    E._WGRAD_WS[0] = torch.empty(1 << 25, dtype=torch.float32)
    return torch.device('cpu')


def forward():
    import torch
    dev = emulate()
    from embodiedscan_amd import engine as E, pipeline
    from embodiedscan_amd.config import build_detector, load_config
    from embodiedscan_amd.synth import make_scan
    from oracle import model as OM
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


def train():
    import torch
    dev = emulate()
    import test_gpu_model as T
    from embodiedscan_amd import pipeline
    from embodiedscan_amd.config import build_detector
    from embodiedscan_amd.synth import make_scan
    det = build_detector(os.path.join(ROOT, T.CFG), device=dev, seed=0).to(dev)
    g = torch.Generator().manual_seed(1)                       # (the GPU test's fixture: non-trivial frozen-BN statistics)
    sd = {k: v.cpu() for k, v in det.state_dict().items()}
    for k in sd:
        if k.startswith('backbone.') and k.endswith('running_var'):
            sd[k] = torch.rand(sd[k].shape, generator=g) + 0.5
        if k.startswith('backbone.') and (k.endswith('running_mean') or k.endswith('bn1.bias') or k.endswith('bn2.bias')):
            sd[k] = torch.randn(sd[k].shape, generator=g) * 0.1
    det.load_state_dict({k: v.to(dev) for k, v in sd.items()})
    scans = [make_scan(s, n_views=2, height=60, width=80, img_size=(64, 64), n_points=2500, n_boxes=5) for s in (11, 12)]
    dscans = [pipeline.upload_scan(s, dev) for s in scans]
    t0 = time.time()
    T.test_train_step_parity((det, scans, dscans, sd))
    print(f'mv-3ddet f32 train step (forward + backward; targets bit exact, every gradient vs the f64-calibrated oracle) under '
          f'emulation: PASSED in {time.time() - t0:.0f} s')


def predict():
    """mode='predict'  end to end: eval-mode norm layers, score / top-k selection, box decoding, multi-class rotated
    NMS -- detections identical to the oracle's (the body of tests/test_gpu_predict.py::test_predict_end_to_end on smaller scans)"""
    import numpy as np
    import torch
    dev = emulate()
    from embodiedscan_amd import engine as E, pipeline
    from embodiedscan_amd.config import build_detector
    from embodiedscan_amd.synth import make_scan
    from oracle import model as OM
    det = build_detector(os.path.join(ROOT, 'configs/mv_3ddet.py'), device=dev, seed=0).to(dev)
    det.bbox_head.test_cfg = dict(nms_pre=300, iou_thr=0.5, score_thr=0.09)
    g = torch.Generator().manual_seed(2)
    sd = {k: v.cpu() for k, v in det.state_dict().items()}
    for k in sd:
        if k.endswith('running_var'):
            sd[k] = torch.rand(sd[k].shape, generator=g) * 0.5 + 0.75
        if k.endswith('running_mean'):
            sd[k] = torch.randn(sd[k].shape, generator=g) * 0.05
    det.load_state_dict({k: v.to(dev) for k, v in sd.items()})
    scans = [make_scan(s, n_views=2, height=60, width=80, img_size=(64, 64), n_points=2500, n_boxes=5) for s in (31, 32)]
    batch = pipeline.make_batch([pipeline.upload_scan(s, dev) for s in scans])
    pts_host = [p.cpu() for p in batch['inputs']['points']]
    t0 = time.time()
    data = det.data_preprocessor(batch, False)
    out = det.forward(data['inputs'], data['data_samples'], mode='predict')
    dt = time.time() - t0
    assert det.training and E.TAPE.enabled
    imgs = torch.stack([OM.preprocess_img(torch.from_numpy(s['img']), [123.675, 116.28, 103.53], [58.395, 57.12, 57.375])
                        for s in scans])
    ref = OM.detector_predict(sd, pts_host, imgs, [s['meta'] for s in scans], nms_pre=300, score_thr=0.09, iou_thr=0.5)
    n_det = 0
    for ds, (rb, rs, rl) in zip(out, ref):
        pr = ds.pred_instances_3d
        print(f'detections: emulated kernels {len(pr.scores_3d)} oracle {len(rs)}')
        assert len(pr.scores_3d) == len(rs)
        np.testing.assert_array_equal(pr.labels_3d.cpu().numpy(), rl.numpy())
        np.testing.assert_allclose(pr.scores_3d.cpu().numpy(), rs.numpy(), rtol=2e-5, atol=1e-7)
        np.testing.assert_allclose(pr.bboxes_3d.tensor.cpu().numpy(), rb.numpy(), rtol=3e-4, atol=2e-4)
        n_det += len(rs)
    print(f'predict (2 scans, {n_det} detections: labels identical, scores 2e-5, boxes 3e-4 vs the oracle) under emulation: PASSED in {dt:.0f} s')


def grounder():
    dev = emulate()
    import test_gpu_grounding as T
    t0 = time.time()
    T.test_grounder_train_step_vs_oracle(dev, 'f32')
    print(f'grounder f32 train step (queries, assignments, logits, losses, gradients vs the oracle) under emulation: PASSED in '
          f'{time.time() - t0:.0f} s')


if __name__ == '__main__':
    {'forward': forward, 'train': train, 'grounder': grounder, 'predict': predict}[sys.argv[1] if len(sys.argv) > 1 else 'forward']()
