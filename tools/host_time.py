"""dev tool: how long does the host take to ISSUE one train step (no sync) vs. the synchronised step time."""
import os, sys, time
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
dscans = [pipeline.upload_scan(make_scan(1234 + i, render_device='cuda:0'), dev) for i in range(4)]
def step():
    return det.train_step(pipeline.make_batch(dscans), optim)
for _ in range(3): step()
torch.cuda.synchronize()
t0 = time.perf_counter(); issue = []
for _ in range(6):
    a = time.perf_counter(); step(); issue.append(time.perf_counter() - a)
torch.cuda.synchronize()
print('issue ms per step', [round(x * 1e3, 1) for x in issue], 'wall ms/step', (time.perf_counter() - t0) / 6 * 1e3)
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); step(); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(28)
