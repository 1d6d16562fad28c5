"""One bf16 train step of EVERY mirrored reference configuration (configs/*.py = the model sections of all six `mv-*` files under
the reference's configs/detection|grounding|occupancy, pinned by tests/test_configs.py) at the shipped model sizes, on a small
synthetic batch: builds through the registry, steps, finite losses, non-zero finite gradients, parameters move."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize('name', ['mv_3ddet.py', 'mv_grounding.py', 'mv_grounding_fcaf.py', 'mv_occ.py'])
def test_shipped_config_builds_and_steps(name):
    from embodiedscan_amd import engine as E, pipeline
    from embodiedscan_amd.config import build_detector, build_optim_wrapper, load_config
    from embodiedscan_amd.synth import make_grounding_sample, make_occ_gt, make_scan
    dev = torch.device('cuda:0')
    cfg = load_config(os.path.join(ROOT, 'configs', name))
    det = build_detector(cfg, device=dev, seed=0).to(dev)
    optim = build_optim_wrapper(cfg)
    kind = cfg['model']['type']
    if kind == 'DenseFusionOccPredictor':
        sc = make_scan(900, n_views=4, augment=False, render_device=str(dev))
        oc = make_occ_gt(sc, seed=0)
        make = lambda: pipeline.make_occ_batch([pipeline.upload_scan(sc, dev)], [oc])
    else:
        scans = [make_scan(900 + i, n_views=4, height=240, width=320, img_size=(256, 256), n_points=30000) for i in range(2)]
        ds = [pipeline.upload_scan(s, dev) for s in scans]
        if kind == 'SparseFeatureFusion3DGrounder':
            anns = [make_grounding_sample(s, seed=i) for i, s in enumerate(scans)]
            make = lambda: pipeline.make_grounding_batch(ds, anns)
        else:
            make = lambda: pipeline.make_batch(ds)
    E.PRECISION[0] = 'bf16'
    try:
        n = det.arena.n_train
        losses = det.train_step(make(), optim)
        torch.cuda.synchronize()
        p0 = det.arena.data[:n].clone()
        g = det.arena.grad[:n]
        assert all(np.isfinite(float(v)) for v in losses.values()), losses
        assert torch.isfinite(g).all() and float(g.abs().max()) > 0
        losses = det.train_step(make(), optim)
        torch.cuda.synchronize()
        assert float((det.arena.data[:n] - p0).abs().max()) > 0
    finally:
        E.PRECISION[0] = 'f32'
    print(f'{name}: {kind} built from the mirrored reference model section, two bf16 train steps, losses ' +
          ', '.join(f'{k} {float(v):.4f}' for k, v in losses.items()))
