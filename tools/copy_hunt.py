"""Which torch operators does one train step dispatch, from where?  (round 5: the mv-3ddet kernel trace shows ~116 rocclr copy kernels
and ~87 fills per step beside the es_* launches -- profiles/r5_single_stream_kernel_stats.txt.)
    python tools/copy_hunt.py mv3ddet|grounding|occupancy [scans_per_step]
runs warm-up steps, then ONE step under a TorchDispatchMode and prints every aten operator that touches device memory grouped by
(operator, call site in embodiedscan_amd/ or bench.py), with call counts and bytes."""
import os
import sys
import traceback
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else 'mv3ddet'
    import torch
    from torch.utils._python_dispatch import TorchDispatchMode
    import bench
    from embodiedscan_amd import engine as E, pipeline
    from embodiedscan_amd.config import build_detector, build_optim_wrapper, load_config
    from embodiedscan_amd.synth import make_grounding_sample, make_occ_gt, make_scan
    dev = torch.device('cuda:0')
    E.PRECISION[0] = 'bf16'
    cfgname = {'grounding': 'mv_grounding.py', 'occupancy': 'mv_occ.py', 'mv3ddet': 'mv_3ddet.py'}[kind]
    cfg = load_config(os.path.join(ROOT, 'configs', cfgname))
    det = build_detector(cfg, device=dev, seed=0).to(dev)
    optim = build_optim_wrapper(cfg)
    nv = 10 if kind == 'occupancy' else 20
    nscan = int(sys.argv[2]) if len(sys.argv) > 2 else {'mv3ddet': 4, 'grounding': 4, 'occupancy': 1}[kind]
    scans = []
    for i in range(nscan):
        sc = make_scan(100 + i, n_views=nv, augment=(kind == 'grounding'), render_device=str(dev))
        if kind == 'grounding':
            a = make_grounding_sample(sc, seed=i)
            sc = dict(sc, text=a['text'], tokens_positive=a['tokens_positive'], gt_boxes=a['gt_boxes'], gt_labels=a['gt_labels'])
        elif kind == 'occupancy':
            oc = make_occ_gt(sc, seed=i)
            sc = dict(sc, gt_occupancy=oc['gt_occupancy'], gt_occupancy_masks=oc['gt_occupancy_masks'])
        scans.append(sc)
    make = {'grounding': pipeline.make_grounding_batch, 'occupancy': pipeline.make_occ_batch, 'mv3ddet': pipeline.make_batch}[kind]
    feeder = bench.Feeder([pipeline.pin_batch(scans)], dev)

    def step():
        out = det.train_step(make(feeder.next()), optim)
        feeder.done()
        return out
    for _ in range(4):
        step()
    torch.cuda.synchronize()
    log = defaultdict(lambda: [0, 0])

    def site():
        for fr in reversed(traceback.extract_stack()[:-3]):
            f = fr.filename
            if ('embodiedscan_amd' in f or f.endswith('bench.py')) and 'copy_hunt' not in f:
                return f'{os.path.relpath(f, ROOT)}:{fr.lineno} {fr.name}'
        return '?'

    class Mode(TorchDispatchMode):
        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            out = func(*args, **(kwargs or {}))
            ts = [t for t in (list(args) + [out]) if isinstance(t, torch.Tensor)]
            if any(t.is_cuda for t in ts):
                name = str(func).replace('aten.', '')
                if not any(s in name for s in ('view', 'reshape', 'detach', 'alias', 'as_strided', 'select', 'slice', 'unsqueeze', 'squeeze',
                                               'expand', 'permute', 'transpose', 't.default', 'unbind', 'split', 'stride', 'size', 'numel',
                                               'is_pinned', 'empty', 'record_stream', '_local_scalar', 'item')):
                    nb = max((t.numel() * t.element_size() for t in ts if t.is_cuda), default=0)
                    e = log[(name, site())]
                    e[0] += 1
                    e[1] += nb
            return out
    with Mode():
        step()
    torch.cuda.synchronize()
    tot = sum(v[0] for v in log.values())
    print(f'{kind}: {tot} device-touching torch operators in one step ({nscan} scans)')
    byop = defaultdict(int)
    for (name, _), v in log.items():
        byop[name] += v[0]
    print('by operator:', sorted(byop.items(), key=lambda kv: -kv[1]))
    for (name, where), v in sorted(log.items(), key=lambda kv: -kv[1][0]):
        print(f'{v[0]:5d}  {v[1] / 1e6:10.3f} MB  {name:32s} {where}')


if __name__ == '__main__':
    main()
