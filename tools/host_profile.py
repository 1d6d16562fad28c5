#!/usr/bin/env python
"""dev tool (round 4): where does the HOST spend its time issuing one mv-3ddet train step?
(1) raw costs: one kernel launch through hip.call, torch.empty on the device, an event record + cross-stream wait;
(2) per step: host time inside train_step() (prefetched batch: no host round trip inside), inside prefetch(), wall time;
(3) cProfile of one train_step sorted by own time."""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench as B
from embodiedscan_amd import engine as E, hip, pipeline
from embodiedscan_amd.config import build_detector, build_optim_wrapper, load_config
from embodiedscan_amd.hip import P, call
from embodiedscan_amd.synth import make_scan

dev = torch.device('cuda:0')
E.PRECISION[0] = 'bf16'
x = torch.zeros(1024, device=dev)
st = torch.cuda.current_stream().cuda_stream
N = 5000
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(N):
    call('es_relu_fwd', P(x), 1024, st)
t1 = time.perf_counter()
torch.cuda.synchronize()
print(f'kernel launch through hip.call: {(t1 - t0) / N * 1e6:.2f} us host each (queue never drained: {(time.perf_counter() - t0) / N * 1e6:.2f} us incl. drain)')
t0 = time.perf_counter()
for _ in range(N):
    y = torch.empty((1000, 64), device=dev)
print(f'torch.empty on the device: {(time.perf_counter() - t0) / N * 1e6:.2f} us')
s2 = torch.cuda.Stream()
ev = torch.cuda.Event()
t0 = time.perf_counter()
for _ in range(N):
    ev.record()
    s2.wait_event(ev)
print(f'event record + wait: {(time.perf_counter() - t0) / N * 1e6:.2f} us')
t0 = time.perf_counter()
for _ in range(N):
    x.zero_()
print(f'torch zero_ (fill kernel): {(time.perf_counter() - t0) / N * 1e6:.2f} us')
torch.cuda.synchronize()

cfg = load_config(os.path.join(ROOT, 'configs', 'mv_3ddet.py'))
det = build_detector(cfg, device=dev, seed=0).to(dev)
optim = build_optim_wrapper(cfg)
scans = [make_scan(1234 + i, n_views=20, render_device=str(dev)) for i in range(12)]
feeder = B.Feeder([pipeline.pin_batch(scans[r * 4:(r + 1) * 4]) for r in range(3)], dev)
make = lambda: pipeline.make_batch(feeder.next())
nxt = None
for it in range(10):                       # reach the steady state (graphs captured, allocator warm)
    batch = nxt if nxt is not None else make()
    det.train_step(batch, optim)
    feeder.done()
    nxt = det.prefetch(make)
torch.cuda.synchronize()
# aligned host / device timeline of 6 steady steps: host perf_counter at the stage boundaries, device events on the main stream
# (step begin / end) and on the prefetch stream (begin / end), all relative to one synchronised origin
base = torch.cuda.Event(enable_timing=True)
base.record()
torch.cuda.synchronize()
h0 = time.perf_counter()
log = []
main = torch.cuda.current_stream()
for it in range(6):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    E.HOST_MARKS = []
    E.MARKS = []
    a = time.perf_counter()
    ev[0].record(main)
    batch = nxt if nxt is not None else make()
    det.train_step(batch, optim)
    feeder.done()
    ev[1].record(main)
    b = time.perf_counter()
    hm, E.HOST_MARKS = E.HOST_MARKS, None
    dm, E.MARKS = E.MARKS, None
    pst = det._pf_stream
    ev[2].record(pst)
    nxt = det.prefetch(make)
    ev[3].record(pst)
    c = time.perf_counter()
    log.append((a - h0, b - h0, c - h0, ev, [(n, t - h0) for n, t in hm], dm))
torch.cuda.synchronize()
for i, (a, b, c, ev, hm, dm) in enumerate(log):
    d = [base.elapsed_time(e) for e in ev]
    print(f'step {i}: HOST train_step {a * 1e3:7.2f} .. {b * 1e3:7.2f} ms, prefetch .. {c * 1e3:7.2f} | DEVICE main stream {d[0]:7.2f} .. {d[1]:7.2f} ms, '
          f'prefetch stream {d[2]:7.2f} .. {d[3]:7.2f}')
    print('        host stage boundaries: ' + ', '.join(f'{n.split()[0]} {t * 1e3:.2f}' for n, t in hm))
    print('        device stage boundaries (main stream): ' + ', '.join(f'{n.split()[0]} {base.elapsed_time(e):.2f}' for n, e in dm))
rows = []

for it in range(4):
    if it == 2:
        pr = cProfile.Profile()
        pr.enable()
    a = time.perf_counter()
    batch = nxt if nxt is not None else make()
    det.train_step(batch, optim)
    feeder.done()
    b = time.perf_counter()
    if it == 2:
        pr.disable()
    nxt = det.prefetch(make)
    c = time.perf_counter()
    rows.append((b - a, c - b))
torch.cuda.synchronize()
for i, (a, b) in enumerate(rows):
    print(f'step {i}: host in train_step {a * 1e3:.2f} ms, in prefetch {b * 1e3:.2f} ms')
pstats.Stats(pr).sort_stats('tottime').print_stats(35)
