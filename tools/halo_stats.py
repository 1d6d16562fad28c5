"""in-situ halo statistics (dev tool): one mv-3ddet train step at the bench's batch; for every halo plan built: rows, halo rows per
256-row tile (mean / p95 / max), tiles beyond the 704 resident rows"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from embodiedscan_amd import engine as E, pipeline
from embodiedscan_amd.config import build_detector, build_optim_wrapper, load_config
from embodiedscan_amd.synth import make_scan

dev = torch.device('cuda:0')
E.PRECISION[0] = 'bf16'
cfg = load_config(os.path.join(ROOT, 'configs', 'mv_3ddet.py'))
det = build_detector(cfg, device=dev, seed=0).to(dev)
optim = build_optim_wrapper(cfg)
scans = [make_scan(1234 + i, render_device='cuda:0') for i in range(4)]
batch = pipeline.make_batch([pipeline.upload_scan(s, dev) for s in scans])
plans = []
orig = E.halo_plan


def spy(nbr):
    fresh = getattr(nbr, '_halo', None) is None
    p = orig(nbr)
    if fresh:
        plans.append((nbr.shape[0], p[2]))
    return p


E.halo_plan = spy
det.train_step(batch, optim)
torch.cuda.synchronize()
for n, hc in plans:
    h = hc.float()
    q = torch.quantile(h, torch.tensor([0.5, 0.95], device=dev))
    print(f'rows {n:7d} tiles {hc.numel():5d} halo mean {float(h.mean()):7.1f} median {float(q[0]):6.0f} p95 {float(q[1]):6.0f} max {int(h.max()):5d} '
          f'tiles > 704: {int((hc > 704).sum()):5d} ({float((hc > 704).float().mean()) * 100:.1f} %) pages total {int(((hc + 703) // 704).clamp(min=1).sum())}')
