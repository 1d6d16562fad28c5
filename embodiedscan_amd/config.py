"""Tiny loader for mmengine-style python configs (`_base_` inheritance + plain dict merge), enough to read
the reference's configs/detection/*.py unchanged (tools/train.py:64 uses mmengine.Config.fromfile)."""
import os


def _merge(base, new):
    for k, v in new.items():
        if isinstance(v, dict) and isinstance(base.get(k), dict) and not v.pop('_delete_', False):
            _merge(base[k], v)
        else:
            base[k] = v
    return base


def load_config(path):
    path = os.path.abspath(path)
    ns = {}
    with open(path) as f:
        exec(compile(f.read(), path, 'exec'), ns)
    cfg = {}
    bases = ns.get('_base_', [])
    if isinstance(bases, str):
        bases = [bases]
    for b in bases:
        _merge(cfg, load_config(os.path.join(os.path.dirname(path), b)))
    own = {k: v for k, v in ns.items() if not k.startswith('_') and not callable(v) and not isinstance(v, type(os))}
    return _merge(cfg, own)


def build_detector(cfg_or_path, device='cuda:0', seed=0):
    from . import models  # noqa: F401  (registers the classes)
    from .registry import MODELS
    cfg = load_config(cfg_or_path) if isinstance(cfg_or_path, str) else cfg_or_path
    return MODELS.build(cfg['model'], device=device, seed=seed)


def build_dataloader(cfg_or_path, split='train', rank=0, world=1, seed=0, num_threads=None, pin=True, workers='thread',
                     **overrides):
    """`{split}_dataloader` section of a reference config -> (EmbodiedScanDataset, ScanLoader).  `RepeatDataset(times=k)`
    becomes the loader's `times`; `num_workers` forked CPU pipelines become decode workers (`workers='thread'` or
    'process': forked workers writing into shared pinned slots; the transforms themselves run on the device);
    `overrides` replace dataset arguments (data_root, ann_file, metainfo, ...).  Defaults follow the reference stack:
    mmengine's DefaultSampler shuffles unless told otherwise and torch's DataLoader keeps the last partial batch."""
    from . import datasets  # noqa: F401  (registers the classes)
    from .datasets import ScanLoader
    from .registry import DATASETS
    cfg = load_config(cfg_or_path) if isinstance(cfg_or_path, str) else cfg_or_path
    dl = cfg[f'{split}_dataloader']
    dcfg, times = dict(dl['dataset']), 1
    if dcfg.get('type', '').split('.')[-1] == 'RepeatDataset':
        times, dcfg = dcfg.get('times', 1), dict(dcfg['dataset'])
    dcfg.update(overrides)
    ds = DATASETS.build(dcfg)
    sampler = dl.get('sampler') or {}
    assert sampler.get('type', 'DefaultSampler').split('.')[-1] == 'DefaultSampler', sampler
    threads = num_threads if num_threads is not None else max(8, 4 * int(dl.get('num_workers', 1)))
    return ds, ScanLoader(ds, batch_size=dl.get('batch_size', 1), rank=rank, world=world, shuffle=sampler.get('shuffle', True),
                          seed=seed, times=times, num_threads=threads, prefetch=2 * threads, pin=pin,
                          drop_last=dl.get('drop_last', False), workers=workers)


def build_optim_wrapper(cfg):
    from .optim import OptimWrapper
    ow = cfg.get('optim_wrapper', {})
    opt = ow.get('optimizer', {})
    assert opt.get('type', 'AdamW') == 'AdamW'
    clip = ow.get('clip_grad') or {}
    pw = {k: v for k, v in (ow.get('paramwise_cfg') or {}).get('custom_keys', {}).items()
          if not k.startswith('text_encoder')}        # the text encoder is a frozen external module, not in the arena
    return OptimWrapper(lr=opt.get('lr', 1e-3), weight_decay=opt.get('weight_decay', 1e-2),
                        betas=opt.get('betas', (0.9, 0.999)), eps=opt.get('eps', 1e-8),
                        max_norm=clip.get('max_norm', 0.0), paramwise=pw)


def build_param_scheduler(cfg, optim):
    """`param_scheduler` of the config (a dict or a list of dicts) -> scheduler objects bound to the optimiser wrapper"""
    from .optim import MultiStepLR
    ps = cfg.get('param_scheduler') or []
    ps = [ps] if isinstance(ps, dict) else list(ps)
    out = []
    base_lr = cfg.get('optim_wrapper', {}).get('optimizer', {}).get('lr', optim.initial_lr)
    for p in ps:
        assert p.get('type') == 'MultiStepLR', f"only MultiStepLR is configured for this path, got {p.get('type')}"
        out.append(MultiStepLR(optim, p['milestones'], p.get('gamma', 0.1), p.get('begin', 0), p.get('end', 10 ** 9),
                               p.get('by_epoch', True), base_lr=base_lr))
    return out
