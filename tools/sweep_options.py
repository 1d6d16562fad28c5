#!/usr/bin/env python
"""dev tool: the bench's mv-3ddet step under different run-time tuning options (es_set_option), in ONE process -- detector
and batches are built once, every variant runs `--warmup` + `--steps` steps.  Prints one line per variant.
  python tools/sweep_options.py [--steps 12]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=12)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--config', default='mv3ddet', choices=['mv3ddet', 'occupancy'])
    ap.add_argument('--variants', default=None,
                    help='semicolon-separated variants, each a comma-separated list of key=value for es_set_option, e.g. '
                         '"10=1;10=2;10=2,8=192": only these (each bracketed by the defaults) instead of the full sweep')
    args = ap.parse_args()
    import torch
    import bench as B
    from embodiedscan_amd import engine as E, hip, pipeline
    from embodiedscan_amd.config import build_detector, build_optim_wrapper, load_config
    from embodiedscan_amd.synth import make_scan
    dev = torch.device('cuda:0')
    E.PRECISION[0] = 'bf16'
    occ = args.config == 'occupancy'
    cfg = load_config(os.path.join(ROOT, 'configs', 'mv_occ.py' if occ else 'mv_3ddet.py'))
    det = build_detector(cfg, device=dev, seed=0).to(dev)
    optim = build_optim_wrapper(cfg)
    if occ:
        from embodiedscan_amd.synth import make_occ_gt
        scans = []
        for i in range(3):
            sc = make_scan(4321 + i, n_views=10, augment=False, render_device=str(dev))
            oc = make_occ_gt(sc, seed=i)
            scans.append(dict(sc, gt_occupancy=oc['gt_occupancy'], gt_occupancy_masks=oc['gt_occupancy_masks']))
        feeder = B.Feeder([pipeline.pin_batch([s]) for s in scans], dev)
        make = pipeline.make_occ_batch
    else:
        scans = [make_scan(1234 + i, n_views=20, render_device=str(dev)) for i in range(12)]
        feeder = B.Feeder([pipeline.pin_batch(scans[r * 4:(r + 1) * 4]) for r in range(3)], dev)
        make = pipeline.make_batch

    def step():
        ds = feeder.next()
        out = det.train_step(make(ds), optim)
        feeder.done()
        return out

    def run(label):
        for _ in range(args.warmup):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / args.steps * 1e3
        print(json.dumps(dict(variant=label, ms_per_step=round(ms, 3))), flush=True)
        return ms
    defaults = {4: 2048, 5: 512, 6: 1024, 7: 256, 8: 384}
    names = {4: 'wgrad big target', 5: 'wgrad big min rows', 6: 'wgrad small target', 7: 'wgrad ws cap MB', 8: 'fwd split wgs'}
    run('default (first)')
    run('default (again)')
    if args.variants:
        base = dict(defaults)
        base.update({10: 2, 11: 768, 12: 0, 14: 1, 15: 4096, 16: 0, 17: 1})
        for var in args.variants.split(';'):
            kv = [tuple(p.split('=')) for p in var.split(',') if p]
            old = {}
            for k, v in kv:
                if k.isdigit():                       # es_set_option key
                    hip.raw('es_set_option')(int(k), int(v))
                else:                                 # engine flag (a one-element list), e.g. NORM_SHADOW=0, ACT16=0
                    old[k] = getattr(E, k)[0]
                    getattr(E, k)[0] = bool(int(v))
            run('options ' + var)
            for k, _ in kv:
                if k.isdigit():
                    hip.raw('es_set_option')(int(k), base[int(k)])
                else:
                    getattr(E, k)[0] = old[k]
            run('default (after ' + var + ')')
        return
    if occ:
        hip.raw('es_set_option')(2, 0)
        run('256 x 256 weight-gradient tile OFF')
        hip.raw('es_set_option')(2, 1)
    for key, values in ((4, (4096, 16384)), (5, (256, 1024, 2048)), (6, (2048, 8192)), (7, (128, 512)), (8, (192, 768))):
        for v in values:
            hip.raw('es_set_option')(key, v)
            run(f'{names[key]} = {v} (default {defaults[key]})')
        hip.raw('es_set_option')(key, defaults[key])
    for flag, lab in ((E.ACT16, 'bf16 activation storage OFF'), (E.WGRAD_OVERWRITE, 'first-write overwrite OFF')):
        flag[0] = False
        run(lab)
        flag[0] = True
    run('default (last)')


if __name__ == '__main__':
    main()
