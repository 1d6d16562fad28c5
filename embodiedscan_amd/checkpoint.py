"""Checkpoint IO in the layout mmengine writes for the reference (`torch.save({'meta', 'state_dict', 'optimizer'})`,
mmengine/runner/checkpoint.py; the released mv-3ddet.pth is such a file, README.md:206): `state_dict` carries the
reference's parameter names and shapes (backbone.layer1.0.conv1.weight (O,I,KH,KW), backbone_3d.conv1.kernel (K,I,O),
bbox_head.conv_cls.kernel ...), so a checkpoint written here loads into the reference and vice versa.  The optimiser
entry is keyed by parameter NAME (see OptimWrapper.state_dict) and is only meant for resuming in this framework."""
import torch


def _arena(model):
    return getattr(model, 'arena', model)


def save_checkpoint(model, path, optim=None, meta=None):
    arena = _arena(model)
    ckpt = dict(meta=dict(meta or {}, framework='embodiedscan_amd', version=2),
                state_dict={k: v.cpu() for k, v in arena.state_dict().items()})
    if optim is not None:
        od = optim.state_dict(arena)
        ckpt['optimizer'] = dict(step=od['step'], param_groups=od['param_groups'],
                                 exp_avg={k: v.cpu() for k, v in od['exp_avg'].items()},
                                 exp_avg_sq={k: v.cpu() for k, v in od['exp_avg_sq'].items()})
    torch.save(ckpt, path)
    return path


def load_checkpoint(model, path, optim=None, strict=False, map_location='cpu'):
    """Returns (missing, unexpected, meta).  Accepts a full mmengine checkpoint or a bare state dict; strips the
    'module.' prefix DistributedDataParallel adds.  After loading, frozen-BN folds and bf16 weight copies are refreshed
    through model.load_state_dict when `model` is a detector."""
    ckpt = torch.load(path, map_location=map_location, weights_only=False)
    sd = ckpt.get('state_dict', ckpt) if isinstance(ckpt, dict) else ckpt
    sd = {(k[7:] if k.startswith('module.') else k): v for k, v in sd.items()}
    arena = _arena(model)
    dev = arena.data.device
    sd = {k: v.to(dev) for k, v in sd.items() if torch.is_tensor(v)}
    # a detector also refreshes its derived state (frozen-BN folds, bf16 weight copies); a bare arena just copies
    missing, unexpected = model.load_state_dict(sd, strict=strict)
    if optim is not None and isinstance(ckpt, dict) and 'optimizer' in ckpt and 'exp_avg' in ckpt['optimizer']:
        od = ckpt['optimizer']
        optim.load_state_dict(arena, dict(step=od['step'], param_groups=od.get('param_groups', [{}]),
                                          exp_avg={k: v.to(dev) for k, v in od['exp_avg'].items()},
                                          exp_avg_sq={k: v.to(dev) for k, v in od['exp_avg_sq'].items()}))
    return missing, unexpected, (ckpt.get('meta', {}) if isinstance(ckpt, dict) else {})
