"""Minimal registry with the reference's plug-in protocol (embodiedscan/registry.py:10-31 on top of
mmengine.Registry): `@MODELS.register_module()` and `MODELS.build(dict(type=..., **kwargs))`, with the
scope prefixes the shipped configs use ('mmdet.ResNet', 'mmdet.FocalLoss', ...)."""


class Registry:
    def __init__(self, name):
        self.name, self._modules = name, {}

    def register_module(self, name=None, module=None):
        def deco(cls):
            self._modules[name or cls.__name__] = cls
            return cls
        if module is not None:
            return deco(module)
        return deco

    def get(self, key):
        if key in self._modules:
            return self._modules[key]
        short = key.split('.')[-1]
        if key.split('.')[0] in ('embodiedscan', 'mmdet3d') and short in self._modules:
            return self._modules[short]
        raise KeyError(f'{key} is not in the {self.name} registry')

    def build(self, cfg, **default_args):
        if cfg is None:
            return None
        cfg = dict(cfg)
        for k, v in default_args.items():
            cfg.setdefault(k, v)
        cls = self.get(cfg.pop('type'))
        return cls(**cfg)


MODELS = Registry('model')
TASK_UTILS = Registry('task util')
DATASETS = Registry('dataset')
TRANSFORMS = Registry('transform')
METRICS = Registry('metric')
