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
