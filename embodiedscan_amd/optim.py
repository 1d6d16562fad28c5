"""Optimiser wrapper over the flat parameter arena (mmengine OptimWrapper + torch AdamW + clip_grad as
configured at configs/detection/mv-det3d_8xb4_embodiedscan-3d-284class-9dof.py:219-223), with the
data-parallel gradient exchange folded in: ONE RCCL all-reduce of the flat gradient buffer per step."""
import os
import struct

import torch
from .hip import P, call
from .parallel import allreduce_mean_

# bf16 mode: AdamW writes the bf16 copies of the convolution kernels in the same pass (es_adamw_table) instead of leaving them to
# the next step's es_cast_weights_table launch (8 B per parameter less; ES_ADAMW_CAST=0: the two-pass path)
ADAMW_CAST = [os.environ.get('ES_ADAMW_CAST', '1') != '0']
AW_CHUNK = 4096             # elements per work item of a plain range (csrc/optim.hip)


class OptimWrapper:
    def __init__(self, lr=1e-3, weight_decay=1e-4, betas=(0.9, 0.999), eps=1e-8, max_norm=10.0, paramwise=None):
        self.lr, self.wd, self.betas, self.eps, self.max_norm = lr, weight_decay, betas, eps, max_norm
        # mmengine paramwise_cfg.custom_keys: {substring of the parameter name: dict(lr_mult=..., decay_mult=...)}
        # (configs/grounding/...py:196-201: decoder lr x0.1).  Longest key wins, as in DefaultOptimWrapperConstructor.
        self.paramwise = dict(paramwise or {})
        self.groups = None
        self.initial_lr = lr
        self.step = 0
        self.m = self.v = None
        self.last_norm = None
        self._table = None          # (key, device table, rows, work items, covered Params, frozen Params) of es_adamw_table
        self.last_path = None       # 'table' | 'flat': which AdamW pass the last step ran (bench.py reports it)

    def state_init(self, arena):
        n = arena.n_train
        self.m = torch.zeros(n, dtype=torch.float32, device=arena.data.device)
        self.v = torch.zeros(n, dtype=torch.float32, device=arena.data.device)
        self.partial = torch.empty(2048 * 64, dtype=torch.float64, device=arena.data.device)     # workspace of es_grad_norm
        self.norm = torch.zeros(1, dtype=torch.float32, device=arena.data.device)

    def _build_groups(self, arena):
        """contiguous ranges of the flat arena sharing (lr_mult, decay_mult)"""
        keys = sorted(self.paramwise, key=lambda k: (-len(k), k))
        groups = []
        for name in arena.trainable_names():
            o, n = arena.offsets[name]
            e = o + (n + 3) // 4 * 4
            lm, dm = 1.0, 1.0
            for k in keys:
                if k in name:
                    lm, dm = self.paramwise[k].get('lr_mult', 1.0), self.paramwise[k].get('decay_mult', 1.0)
                    break
            if groups and groups[-1][1] == o and groups[-1][2:] == (lm, dm):
                groups[-1] = (groups[-1][0], e, lm, dm)
            else:
                groups.append((o, e, lm, dm))
        self.groups = groups

    def _adamw_table(self, arena):
        """_adamw_table_build with BOTH outcomes cached per (cast table, arena, groups): a negative answer (aliased / strided
        kernels: the two-pass path) used to rebuild the whole table on every step"""
        from . import engine as E
        tab = E._CAST_TABLE.get(arena.data.device) if (ADAMW_CAST[0] and E.PRECISION[0] == 'bf16' and arena.data.is_cuda) else None
        if tab is None:
            return None
        key = (id(tab), arena.data.data_ptr(), arena.n_train, id(self.groups))
        if getattr(self, '_table_none', None) == key:
            return None
        res = self._adamw_table_build(arena)
        if res is None:
            self._table_none = (id(tab), arena.data.data_ptr(), arena.n_train, id(self.groups))     # (groups may have been built inside)
        return res

    def _adamw_table_build(self, arena):
        """rows of es_adamw_table for this arena, or None when the one-pass update does not apply: every trainable convolution
        kernel registered with the engine's cast table that lives in this arena becomes a tile row, the rest of the trainable
        arena (norm parameters, biases, padding) plain ranges, each with the (lr_mult, decay_mult) of its paramwise group"""
        from . import engine as E
        dev = arena.data.device
        tab = E._CAST_TABLE.get(dev) if (ADAMW_CAST[0] and E.PRECISION[0] == 'bf16' and arena.data.is_cuda) else None
        if tab is None:
            return None
        key = (id(tab), arena.data.data_ptr(), arena.n_train, id(self.groups))
        if self._table is not None and self._table[0] == key:
            return self._table
        n, base = arena.n_train, arena.data.data_ptr()
        if self.paramwise and self.groups is None:
            self._build_groups(arena)
        groups = list(self.groups) if self.paramwise else [(0, n, 1.0, 1.0)]
        if self.paramwise:                                      # (alignment padding between groups, if any, keeps multipliers 1)
            full, pos = [], 0
            for a, b, lm, dm in groups:
                if a > pos:
                    full.append((pos, a, 1.0, 1.0))
                full.append((a, b, lm, dm))
                pos = b
            if pos < n:
                full.append((pos, n, 1.0, 1.0))
            groups = full
        convs, frozen, seen = [], [], set()
        for p in tab['params']:
            off = (p.d.data_ptr() - base) // 4
            inside = base <= p.d.data_ptr() < base + 4 * arena.data.numel()
            if not inside:
                continue
            if off >= n:
                frozen.append(p)
                continue
            if off in seen or not p.d.is_contiguous() or off + p.d.numel() > n:
                return None                                     # aliased / strided kernels: leave them to the two-pass path
            seen.add(off)
            convs.append((off, p))
        convs.sort(key=lambda t: t[0])
        for (o0, p0), (o1, _) in zip(convs, convs[1:]):
            if o0 + p0.d.numel() > o1:
                return None
        dbits = lambda x: struct.unpack('<q', struct.pack('<d', float(x)))[0]
        rows, items, ci = [], 0, 0
        for a, b, lm, dm in groups:
            pos = a
            while pos < b:
                if ci < len(convs) and convs[ci][0] < b and convs[ci][0] >= pos:
                    off, p = convs[ci]
                    if off > pos:
                        rows.append([pos, off - pos, 0, 0, 0, 0, items, dbits(lm), dbits(dm)])
                        items += (off - pos + AW_CHUNK - 1) // AW_CHUNK
                    K, A, B = p.d.shape
                    if off + p.d.numel() > b:
                        return None                             # a kernel straddling two groups cannot happen (groups are whole tensors)
                    rows.append([off, K, A, B, p.bf_n.data_ptr(), p.bf_t.data_ptr(), items, dbits(lm), dbits(dm)])
                    items += K * ((A + 63) // 64) * ((B + 63) // 64)
                    pos = off + p.d.numel()
                    ci += 1
                else:
                    rows.append([pos, b - pos, 0, 0, 0, 0, items, dbits(lm), dbits(dm)])
                    items += (b - pos + AW_CHUNK - 1) // AW_CHUNK
                    pos = b
        if ci != len(convs) or items >= (1 << 31):
            return None
        as_i64 = lambda r: [v - (1 << 64) if v >= (1 << 63) else v for v in r]
        t = torch.tensor([as_i64(r) for r in rows], dtype=torch.int64).to(dev)
        self._table = (key, t, len(rows), items, [p for _, p in convs], frozen, tab)
        return self._table

    def state_dict(self, arena):
        """Resume state: AdamW moments keyed by the reference's parameter names (reference shapes), the step count
        and the hyper-parameters.  (torch.optim's own state_dict is keyed by parameter index in module registration
        order, which this framework does not reproduce; the names are what both sides share.)"""
        if self.m is None:
            self.state_init(arena)
        return dict(step=self.step, exp_avg=arena.flat_to_ref(self.m), exp_avg_sq=arena.flat_to_ref(self.v),
                    param_groups=[dict(lr=self.lr, weight_decay=self.wd, betas=tuple(self.betas), eps=self.eps,
                                       max_norm=self.max_norm)])

    def load_state_dict(self, arena, sd):
        if self.m is None:
            self.state_init(arena)
        self.step = int(sd['step'])
        arena.ref_to_flat(sd['exp_avg'], self.m)
        arena.ref_to_flat(sd['exp_avg_sq'], self.v)
        g = sd.get('param_groups', [{}])[0]
        self.lr, self.wd = g.get('lr', self.lr), g.get('weight_decay', self.wd)
        self.betas, self.eps = tuple(g.get('betas', self.betas)), g.get('eps', self.eps)
        self.max_norm = g.get('max_norm', self.max_norm)

    def update_params(self, arena):
        if self.m is None:
            self.state_init(arena)
        s = torch.cuda.current_stream().cuda_stream
        reducer = getattr(arena, 'reducer', None)
        n = arena.n_train
        self.step += 1
        gscale = 1.0
        if reducer is not None and not reducer.use_sumsq and arena.grad.is_cuda:
            # clip norm under the bucket all-reduces: every reduced chunk's sum of squares is taken on the reducer's side
            # stream right behind its collective, into the REDUCER's buffer (takes effect from the next step's launches on;
            # any number of OptimWrappers may come and go on one detector)
            reducer.use_sumsq = True
        chunks = reducer.finish() if reducer is not None else 0   # buckets launched during backward (parallel.py) ...
        if chunks and reducer.last_sumsq_ok:
            import torch.distributed as dist
            from .parallel import SUMSQ_BLOCK
            gscale = 1.0 / dist.get_world_size()              # the arena holds the SUM over ranks: mean folded into AdamW
            call('es_norm_from_partials', P(reducer.partial), SUMSQ_BLOCK * chunks, gscale, P(self.norm), s)
        else:
            if chunks:
                reducer.scale_()
            else:
                allreduce_mean_(arena.grad)                   # ... else one flat all-reduce (RCCL over xGMI)
            call('es_grad_norm', P(arena.grad), n, P(self.partial), P(self.norm), s)
        from . import engine
        table = self._adamw_table(arena)
        if table is not None:
            # one pass: AdamW + the bf16 copies of every trainable kernel; copies that were up to date stay so (frozen kernels:
            # weights untouched), so the next step's refresh_weight_copies() launches nothing
            _, tdev, nrows, items, convs, frozen, tab = table
            call('es_adamw_table', P(arena.data), P(arena.grad), P(self.m), P(self.v), P(tdev), nrows, items, float(self.lr),
                 float(self.betas[0]), float(self.betas[1]), float(self.eps), float(self.wd), self.step,
                 float(self.max_norm if self.max_norm else 0.0), P(self.norm), gscale, s)
            old = engine.WEIGHT_VERSION[0]
            engine.WEIGHT_VERSION[0] += 1
            for p in convs:
                p.bf_step = engine.WEIGHT_VERSION[0]
            for p in frozen:
                if p.bf_step == old:
                    p.bf_step = engine.WEIGHT_VERSION[0]
            self.last_norm, self.last_gscale, self.last_path = self.norm, gscale, 'table'
            return
        self.last_path = 'flat'
        if not self.paramwise:
            call('es_adamw_step', P(arena.data), P(arena.grad), P(self.m), P(self.v), n, float(self.lr),
                 float(self.betas[0]), float(self.betas[1]), float(self.eps), float(self.wd), self.step,
                 float(self.max_norm if self.max_norm else 0.0), P(self.norm), gscale, s)
        else:
            if self.groups is None:
                self._build_groups(arena)
            for a, b, lm, dm in self.groups:           # one launch per (lr_mult, decay_mult) range; the clip norm is global
                call('es_adamw_step', arena.data.data_ptr() + 4 * a, arena.grad.data_ptr() + 4 * a, self.m.data_ptr() + 4 * a,
                     self.v.data_ptr() + 4 * a, b - a, float(self.lr * lm), float(self.betas[0]), float(self.betas[1]),
                     float(self.eps), float(self.wd * dm), self.step, float(self.max_norm if self.max_norm else 0.0),
                     P(self.norm), gscale, s)
        self.last_norm = self.norm
        self.last_gscale = gscale               # < 1: the arena still holds the SUM over ranks (mean folded into the AdamW pass)
        from . import engine
        engine.WEIGHT_VERSION[0] += 1          # bf16 weight copies are stale now


class MultiStepLR:
    """mmengine MultiStepLR(by_epoch=True) as configured at configs/detection/mv-det3d_...py:225-230: the learning rate
    of the wrapper is base_lr * gamma ** (number of milestones already passed), inside [begin, end) epochs."""

    def __init__(self, optim, milestones, gamma=0.1, begin=0, end=10 ** 9, by_epoch=True, base_lr=None):
        assert by_epoch, 'the shipped configs schedule by epoch'
        self.optim, self.milestones, self.gamma = optim, sorted(int(m) for m in milestones), float(gamma)
        self.begin, self.end = int(begin), int(end)
        # the INITIAL learning rate (mmengine keeps it as param_group['initial_lr']): taken from the config, never from
        # the live optim.lr, which after a resume is already decayed
        self.base_lr = float(base_lr if base_lr is not None else getattr(optim, 'initial_lr', optim.lr))
        self.epoch = 0

    def lr_at(self, epoch):
        """mmengine counts a scheduler's steps from its `begin` (_ParamScheduler.step only runs inside [begin, end) and
        `last_step` starts at 0 there): a milestone m is passed after m epochs SINCE begin"""
        # ... and the last step it takes is the one at global epoch end - 1 (it does not run at `end`), so a milestone equal to
        # end - begin is never reached (round-3 advisor)
        e = min(max(epoch, self.begin), self.end - 1) - self.begin
        return self.base_lr * self.gamma ** sum(1 for m in self.milestones if m <= e)

    def step(self):
        """call once after every training epoch (ParamSchedulerHook.after_train_epoch)"""
        self.epoch += 1
        self.optim.lr = self.lr_at(self.epoch)
        return self.optim.lr

    def state_dict(self):
        return dict(epoch=self.epoch, base_lr=self.base_lr)

    def load_state_dict(self, sd):
        self.epoch, self.base_lr = int(sd['epoch']), float(sd['base_lr'])
        self.optim.lr = self.lr_at(self.epoch)
