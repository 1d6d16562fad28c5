"""Next-batch prefetch (detector.prefetch, round 4): the weight-independent prefix of step i+1 (A1-A4, A6 maps and unions,
A18) issued on a side stream under step i's backward must not change ANYTHING the step computes.  Two identically seeded
detectors train for three steps on three distinct batches, one serially, one with the software pipeline bench.py uses:
losses of every step, all gradients of the last step and the parameters after it must be bit-identical (the HIP path is
run-to-run deterministic: tests/test_gpu_config2.py::test_run_to_run...)."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(pipelined, dev, steps=4):
    from embodiedscan_amd import engine as E, pipeline
    from embodiedscan_amd.config import build_detector, build_optim_wrapper, load_config
    from embodiedscan_amd.synth import make_scan
    cfg = load_config(os.path.join(ROOT, 'configs', 'mv_3ddet.py'))
    det = build_detector(cfg, device=dev, seed=0).to(dev)
    optim = build_optim_wrapper(cfg)
    batches = [[pipeline.upload_scan(make_scan(50 + 2 * i + j, n_views=3, height=240, width=320, img_size=(192, 192),
                                               n_points=15000), dev) for j in range(2)] for i in range(3)]
    it = iter(range(10 ** 6))
    make = lambda: pipeline.make_batch(batches[next(it) % 3])
    E.PRECISION[0] = 'bf16'
    losses = []
    try:
        E.WEIGHT_VERSION[0] += 1
        nxt = None
        for s in range(steps):
            batch = nxt if nxt is not None else make()
            out = det.train_step(batch, optim)
            losses.append({k: float(v) for k, v in out.items()})
            nxt = det.prefetch(make) if (pipelined and s + 1 < steps) else None
        torch.cuda.synchronize()
    finally:
        E.PRECISION[0] = 'f32'
    n = det.arena.n_train
    return losses, det.arena.grad[:n].clone(), det.arena.data[:n].clone(), (det._pf_hold, det._prefetched)


def test_prefetched_steps_equal_serial_steps():
    dev = torch.device('cuda:0')
    l0, g0, p0, _ = _run(False, dev)
    l1, g1, p1, (hold, left) = _run(True, dev)
    for a, b in zip(l0, l1):
        assert a == b, (a, b)
    assert torch.equal(g0, g1), float((g0 - g1).abs().max())
    assert torch.equal(p0, p1), float((p0 - p1).abs().max())
    assert left is None and 1 <= len(hold) <= 2          # every prefetched batch was consumed; at most two still pinned
    print(f'{len(l1)} pipelined steps: losses, gradients and parameters bit-identical to the serial steps; '
          f'last losses {l1[-1]}')


def test_train_step_on_other_data_ignores_the_prefetched_batch():
    """prefetch() is advisory: a train_step on a different batch must run the serial path and leave no stale state"""
    from embodiedscan_amd import engine as E, pipeline
    from embodiedscan_amd.config import build_detector, build_optim_wrapper, load_config
    from embodiedscan_amd.synth import make_scan
    dev = torch.device('cuda:0')
    cfg = load_config(os.path.join(ROOT, 'configs', 'mv_3ddet.py'))
    det = build_detector(cfg, device=dev, seed=0).to(dev)
    optim = build_optim_wrapper(cfg)
    ds = [pipeline.upload_scan(make_scan(70 + j, n_views=2, height=120, width=160, img_size=(128, 128), n_points=6000), dev)
          for j in range(2)]
    det.prefetch(lambda: pipeline.make_batch(ds[:1]))
    out = det.train_step(pipeline.make_batch(ds[1:]), optim)      # not the prefetched object
    torch.cuda.synchronize()
    assert det._prefetched is None and det._pf_cur is None and not det._pf_hold
    assert all(torch.isfinite(v) for v in out.values())
