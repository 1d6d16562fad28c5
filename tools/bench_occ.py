#!/usr/bin/env python
"""Occupancy train step (BASELINE config 5: DenseFusionOccPredictor, 10 train views 480x640, 100k points, 40x40x16 volume,
neck 768 -> 1536 -> 3072, 81 classes, batch 1) on one MI355X: ms/step, scans/s and the MFMA fraction of the convolution
engine on the dense 3-D neck (the one MFMA-bound block of the suite, ~4 TFLOP forward / ~12 TFLOP fwd+bwd).
Prints one JSON line.   python tools/bench_occ.py [--steps 5 --warmup 2 --views 10 --precision bf16]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--views', type=int, default=10)
    ap.add_argument('--precision', default='bf16', choices=['bf16', 'f32'])
    args = ap.parse_args()
    import torch
    import bench as B
    from embodiedscan_amd import engine as E, hip, pipeline
    from embodiedscan_amd.config import build_detector, build_optim_wrapper, load_config
    from embodiedscan_amd.synth import make_occ_gt, make_scan
    dev = torch.device('cuda:0')
    E.PRECISION[0] = args.precision
    cfg = load_config(os.path.join(ROOT, 'configs', 'mv_occ.py'))
    det = build_detector(cfg, device=dev, seed=0).to(dev)
    optim = build_optim_wrapper(cfg)
    scans = [make_scan(4321 + i, n_views=args.views, augment=False, render_device=str(dev)) for i in range(2)]
    occs = [make_occ_gt(s, seed=i) for i, s in enumerate(scans)]
    dscans = [pipeline.upload_scan(s, dev) for s in scans]
    state = dict(i=0)

    def step():
        i = state['i'] % len(scans)
        state['i'] += 1
        return det.train_step(pipeline.make_occ_batch([dscans[i]], [occs[i]]), optim)

    for _ in range(args.warmup):
        losses = step()
    torch.cuda.synchronize()
    prof = {'names': B.ENGINE, 'records': [], 'event': lambda: torch.cuda.Event(enable_timing=True)}
    t0 = time.perf_counter()
    for it in range(args.steps):
        hip.PROFILE = prof if it == args.steps - 1 else None
        E.MARKS = [] if it == args.steps - 1 else None
        losses = step()
    recs = B.resolve_pairs(hip, prof['records'])
    marks, E.MARKS = E.MARKS, None
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    hip.PROFILE = None
    peak = B.K_PEAK_MFMA[args.precision]
    eng = B.engine_totals(recs, peak)
    # the dense-neck launches: dense maps have every interior neighbour, channel widths >= 768 on either side
    neck = [r for r in recs if max(B.engine_args(r[0], r[3])[4:6]) >= 768]
    nk = B.engine_totals(neck, peak)
    stages = {}
    for (n0, ev0), (n1, ev1) in zip(marks[:-1], marks[1:]):
        stages[n1] = round(stages.get(n1, 0.0) + ev0.elapsed_time(ev1), 3)
    out = dict(metric='scans/sec (train step) occupancy, 10x(480x640) RGB-D views, 40x40x16 volume', value=round(args.steps / dt, 4),
               unit='scans/s', n_gpus=1, steps=args.steps, warmup=args.warmup, ms_per_step=round(dt / args.steps * 1e3, 3),
               dtype=args.precision, data='synthetic',
               config=dict(workload='DenseFusionOccPredictor: ResNet-50 + FPN, MinkResNet34, IndoorImVoxelNeck 768-1536-3072, '
                                    'ImVoxelOccHead 81 classes, batch 1, full train step incl. AdamW (751 M parameters)', views=args.views),
               losses={k: round(float(v), 6) for k, v in losses.items()},
               roofline=dict(bound='mfma', achieved=nk['tflops'], peak=peak, unit='TFLOP/s', frac=round(nk['tflops'] / peak, 4),
                             kernel='convolution engine on the dense 3-D neck (launches with >= 768 channels)',
                             launches=nk['launches'], kernel_ms=nk['ms'], frac_of_binding_roof=nk['frac_binding'],
                             traffic=None),
               engine_all=dict(launches=eng['launches'], kernel_ms=eng['ms'], tflops=eng['tflops'],
                               frac_of_binding_roof=eng['frac_binding']),
               stage_ms=stages)
    print(json.dumps(out))


if __name__ == '__main__':
    main()
