"""What every detector of this package shares with mmengine's BaseModel protocol: parameter arena binding, reference-named
state dicts, `forward(inputs, data_samples, mode)` dispatch and `train_step(data, optim_wrapper)`
(preprocess -> forward(mode='loss') -> backward over the tape -> optimiser update)."""
import os

import torch
from ... import engine as E
from ... import hip
from ...parallel import BucketedGradReducer, is_dist
from ...params import ParamArena


class DetectorBase:
    _version = 2

    def _init_base(self, specs, device, seed, data_preprocessor):
        from ...registry import MODELS
        self.device = torch.device(device)
        self.data_preprocessor = MODELS.build(data_preprocessor, device=self.device) if data_preprocessor else None
        self.arena = ParamArena(specs, seed=seed)
        self.training = True
        self._bound = False

    def _children(self):
        """[(module, arena prefix)] -- every module with a bind(arena, prefix)"""
        raise NotImplementedError

    def to(self, device):
        self.device = torch.device(device)
        self.arena.to(self.device)
        if self.data_preprocessor is not None:
            self.data_preprocessor.to(self.device)
        self._bound = False
        return self

    def _bind(self):
        if not self._bound:
            if self.arena.data.device != self.device:
                self.arena.to(self.device)
            E.begin_bind(id(self))
            try:
                for m, prefix in self._children():
                    m.bind(self.arena, prefix)
            finally:
                E.end_bind()
            self._bound = True

    def __del__(self):
        try:
            E.release(id(self))
        except Exception:
            pass

    def state_dict(self):
        return self.arena.state_dict()

    def load_state_dict(self, sd, strict=False, tap_order=None):
        res = self.arena.load_state_dict(sd, strict=strict, tap_order=tap_order)
        E.WEIGHT_VERSION[0] += 1
        if self._bound:
            for m, _ in self._children():
                if hasattr(m, 'refresh'):
                    m.refresh()
        return res

    def train(self, mode=True):
        self.training = mode
        for m, _ in self._children():
            if hasattr(m, 'training'):
                m.training = mode
        return self

    def forward(self, inputs, data_samples=None, mode='tensor', **kwargs):
        hip.refresh_stream()
        if mode == 'loss':
            return self.loss(inputs, data_samples, **kwargs)
        elif mode == 'predict':
            return self.predict(inputs, data_samples, **kwargs)
        raise RuntimeError(f'Invalid mode "{mode}". Only supports loss, predict and tensor mode')

    __call__ = forward

    def _predict_guard(self):
        """context: eval mode + tape off"""
        det = self

        class _G:
            def __enter__(self):
                self.was, self.prev = det.training, E.TAPE.enabled
                det.train(False)
                E.TAPE.enabled = False

            def __exit__(self, *exc):
                E.TAPE.enabled = self.prev
                det.train(self.was)
        return _G()

    # data-parallel gradient buckets (parallel.BucketedGradReducer): name-prefix groups + the implicit last part; a
    # detector's forward records (tape index, part) pairs in `_tape_parts`: when the reverse replay of the tape has passed
    # the index, every gradient of that part is complete and its all-reduce starts under the rest of the backward pass
    _bucket_groups = (('backbone.',), ('backbone_3d.',))

    def tape_part(self, part):
        """called in forward: everything recorded on the tape from here on belongs to gradient part `part` (or later ones)"""
        if getattr(self, '_tape_parts', None) is not None:
            self._tape_parts.append((len(E.TAPE.fns), part))

    def _backward(self, red):
        fns = E.TAPE.fns
        hi = len(fns)
        done = set()
        for idx, part in sorted(getattr(self, '_tape_parts', None) or [], reverse=True):
            for fn in reversed(fns[idx:hi]):
                fn()
            hi = idx
            if red is not None:
                E.join_wgrad_streams(final=False)
                red.launch(part)
                done.add(part)
        for fn in reversed(fns[:hi]):
            fn()
        E.join_wgrad_streams()
        if red is not None:
            for part in range(len(red.parts)):
                if part not in done:
                    red.launch(part)
        E.TAPE.clear()

    def train_step(self, data, optim_wrapper):
        E.settle_gc(self)
        E.TAPE.clear()
        hip.refresh_stream()
        E.mark('data (caller)')
        if self.data_preprocessor is not None:
            data = self.data_preprocessor(data, True)
        self._bind()
        self.arena.grad.zero_()
        E.new_grad_epoch()                       # first weight-gradient launch per weight overwrites, later ones add
        self._tape_parts = []
        losses = self.forward(data['inputs'], data['data_samples'], mode='loss')
        E.mark('forward + losses')
        red = None
        if is_dist():
            if getattr(self.arena, 'reducer', None) is None:
                self.arena.reducer = BucketedGradReducer(self.arena, groups=self._bucket_groups)
            red = self.arena.reducer
        self._backward(red)
        E.mark('backward')
        optim_wrapper.update_params(self.arena)
        E.mark('all-reduce + clip + AdamW')
        return losses
