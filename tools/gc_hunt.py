"""Which reference cycles does a train step leave behind?  (round 5: the grounding step's 3.5x outlier every ~20 steps is Python's
generation-2 garbage collection freeing ~100 MB / step of tensors that are only reachable through cycles -- profiles/r5b_*.)
    python tools/gc_hunt.py grounding|occupancy|mv3ddet
runs a few steps with automatic collection off, then collects with DEBUG_SAVEALL and prints the garbage by type, the tensors in it,
and for the largest tensors the chain of garbage referrers."""
import gc
import os
import sys
from collections import Counter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else 'grounding'
    import torch
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
    scans = []
    for i in range(2):
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
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    gc.collect()
    gc.disable()
    a0 = torch.cuda.memory_allocated()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    a1 = torch.cuda.memory_allocated()
    print(f'{kind}: allocated grew {(a1 - a0) / 2 ** 20:.1f} MB over 3 steps with the collector off')
    gc.set_debug(gc.DEBUG_SAVEALL)
    n = gc.collect()
    garbage = list(gc.garbage)
    gc.set_debug(0)
    print(f'collector found {n} unreachable objects; by type:')
    for t, c in Counter(type(o).__name__ for o in garbage).most_common(25):
        print(f'  {c:7d} {t}')
    ids = {id(o): o for o in garbage}
    tens = sorted([o for o in garbage if isinstance(o, torch.Tensor)], key=lambda t: -t.numel() * t.element_size())
    print(f'{len(tens)} tensors in cycles, {sum(t.numel() * t.element_size() for t in tens) / 2 ** 20:.1f} MB')

    def describe(o):
        s = type(o).__name__
        if isinstance(o, torch.Tensor):
            return f'Tensor{tuple(o.shape)}'
        if hasattr(o, '__qualname__'):
            return f'{s}:{o.__qualname__}'
        if s == 'cell':
            try:
                return f'cell->{type(o.cell_contents).__name__}'
            except ValueError:
                return 'cell(empty)'
        if s == 'frame':
            return f'frame:{o.f_code.co_name}'
        if isinstance(o, dict):
            return 'dict{' + ','.join(str(k) for k in list(o)[:6]) + '}'
        return s
    seen_chains = Counter()
    for t in tens[:40]:
        chain, cur = [describe(t)], t
        for _ in range(7):
            refs = [r for r in gc.get_referrers(cur) if id(r) in ids and r is not garbage]
            if not refs:
                break
            cur = refs[0]
            chain.append(describe(cur))
        seen_chains[' <- '.join(chain[1:])] += 1
    for c, k in seen_chains.most_common(12):
        print(f'  x{k}: {c}')
    # functions in cycles
    for t, c in Counter(o.__qualname__ for o in garbage if type(o).__name__ == 'function').most_common(20):
        print(f'  fn {c:5d} {t}')


if __name__ == '__main__':
    main()
