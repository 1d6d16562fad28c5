"""Checkpoint IO in the layout mmengine writes for the reference (`torch.save({'meta', 'state_dict', 'optimizer'})`,
mmengine/runner/checkpoint.py; the released mv-3ddet.pth is such a file, README.md:206): `state_dict` carries the
reference's parameter names and shapes (backbone.layer1.0.conv1.weight (O,I,KH,KW), backbone_3d.conv1.kernel (K,I,O),
bbox_head.conv_cls.kernel ...; BatchNorm `num_batches_tracked` buffers are emitted too), so `load_from=` / a strict
`load_state_dict` on the reference side accepts a checkpoint written here and vice versa.  Resume state of THIS framework
lives under its own keys -- `optimizer_flat` (AdamW moments keyed by parameter NAME, see OptimWrapper.state_dict; the
mmengine key `optimizer` is deliberately not used because torch.optim.AdamW.load_state_dict could not read it) and
`param_schedulers` (mmengine's key: one state dict per scheduler) -- so a resumed run continues the LR schedule."""
import torch


def _arena(model):
    return getattr(model, 'arena', model)


def save_checkpoint(model, path, optim=None, meta=None, schedulers=None):
    arena = _arena(model)
    # a detector's own state_dict() may carry more than the arena (the grounder's frozen `text_encoder.*`)
    src = model.state_dict() if (model is not arena and hasattr(model, 'state_dict')) else arena.state_dict()
    sd = {k: v.cpu() for k, v in src.items()}
    steps = int(optim.step) if optim is not None else 0
    for k in list(sd):                       # nn.BatchNorm buffers a strict reference-side load expects
        if k.endswith('.running_var'):
            sd[k[:-len('running_var')] + 'num_batches_tracked'] = torch.tensor(
                0 if k.startswith('backbone.') else steps, dtype=torch.long)
    ckpt = dict(meta=dict(meta or {}, framework='embodiedscan_amd', version=2), state_dict=sd)
    if optim is not None:
        od = optim.state_dict(arena)
        ckpt['optimizer_flat'] = dict(step=od['step'], param_groups=od['param_groups'],
                                      exp_avg={k: v.cpu() for k, v in od['exp_avg'].items()},
                                      exp_avg_sq={k: v.cpu() for k, v in od['exp_avg_sq'].items()})
    if schedulers:
        ckpt['param_schedulers'] = [s.state_dict() for s in schedulers]
    torch.save(ckpt, path)
    return path


def load_checkpoint(model, path, optim=None, strict=False, map_location='cpu', schedulers=None, tap_order=None):
    """Returns (missing, unexpected, meta).  Accepts a full mmengine checkpoint or a bare state dict; strips the
    'module.' prefix DistributedDataParallel adds.  After loading, frozen-BN folds and bf16 weight copies are refreshed
    through model.load_state_dict when `model` is a detector."""
    ckpt = torch.load(path, map_location=map_location, weights_only=False)
    sd = ckpt.get('state_dict', ckpt) if isinstance(ckpt, dict) else ckpt
    sd = {(k[7:] if k.startswith('module.') else k): v for k, v in sd.items()}
    arena = _arena(model)
    dev = arena.data.device
    sd = {k: v.to(dev) for k, v in sd.items() if torch.is_tensor(v) and not k.endswith('.num_batches_tracked')}
    # a detector also refreshes its derived state (frozen-BN folds, bf16 weight copies); a bare arena just copies
    # tap_order: numbering of the sparse kernels' offsets in the file (ParamArena.tap_permutation); the released reference
    # checkpoints were written by MinkowskiEngine, whose order cannot be verified offline
    missing, unexpected = model.load_state_dict(sd, strict=strict, tap_order=tap_order) if tap_order is not None \
        else model.load_state_dict(sd, strict=strict)
    okey = 'optimizer_flat' if isinstance(ckpt, dict) and 'optimizer_flat' in ckpt else 'optimizer'   # 'optimizer': round-1 files
    if optim is not None and isinstance(ckpt, dict) and okey in ckpt and 'exp_avg' in ckpt[okey]:
        od = ckpt[okey]
        optim.load_state_dict(arena, dict(step=od['step'], param_groups=od.get('param_groups', [{}]),
                                          exp_avg={k: v.to(dev) for k, v in od['exp_avg'].items()},
                                          exp_avg_sq={k: v.to(dev) for k, v in od['exp_avg_sq'].items()}))
    if schedulers and isinstance(ckpt, dict) and 'param_schedulers' in ckpt:
        # after the optimiser (whose restored lr is the already-decayed one): each scheduler re-derives the lr from ITS
        # base_lr and epoch, so neither a double decay nor a reset to the initial lr can happen on resume
        for sch, st in zip(schedulers, ckpt['param_schedulers']):
            sch.load_state_dict(st)
    return missing, unexpected, (ckpt.get('meta', {}) if isinstance(ckpt, dict) else {})
