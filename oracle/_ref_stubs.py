"""Thin stand-ins for the third-party modules the reference imports but which are
absent here (mmengine, mmcv, mmdet, pytorch3d, MinkowskiEngine is left missing),
used ONLY by oracle/make_golden.py to import the reference's pure-PyTorch
functions from /root/reference and record golden vectors.  TEST INFRASTRUCTURE.

Everything is an inert "magic" object except pytorch3d.transforms.{euler_angles_to_matrix,
matrix_to_euler_angles}, restated from pytorch3d v0.7.2 (rotation_conversions.py).
"""
import importlib.abc
import importlib.machinery
import sys
import types
import torch

_PREFIXES = ('mmengine', 'mmcv', 'mmdet', 'mmdet3d', 'pytorch3d', 'open3d', 'cv2', 'terminaltables', 'lmdb', 'mmeval')


class _MagicMeta(type):
    def __getattr__(cls, name):
        if name.startswith('__'):
            raise AttributeError(name)
        return _make(name)

    def __call__(cls, *a, **k):
        if cls.__dict__.get('_pure_magic', False) and len(a) == 1 and not k and callable(a[0]) \
                and not isinstance(a[0], _Magic):
            return a[0]                       # used as a bare decorator
        return super().__call__(*a, **k)


class _Magic(metaclass=_MagicMeta):
    _pure_magic = True

    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        if len(a) == 1 and not k and callable(a[0]) and not isinstance(a[0], _Magic):
            return a[0]                       # @REG.register_module() style
        return _Magic()

    def __getattr__(self, name):
        if name.startswith('__'):
            raise AttributeError(name)
        return _Magic()

    def __iter__(self):
        return iter(())

    def __mro_entries__(self, bases):
        return (_Magic,)


def _make(name):
    return _MagicMeta(name, (_Magic,), {'_pure_magic': True})


class _StubModule(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith('__'):
            raise AttributeError(name)
        v = _make(name)
        setattr(self, name, v)
        return v


class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path, target=None):
        if fullname.split('.')[0] in _PREFIXES:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _StubModule(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        _populate(module)


# --- pytorch3d.transforms restatement (v0.7.2 rotation_conversions.py) ---------------
def _axis_angle_rotation(axis, angle):
    cos, sin = torch.cos(angle), torch.sin(angle)
    one, zero = torch.ones_like(angle), torch.zeros_like(angle)
    if axis == 'X':
        flat = (one, zero, zero, zero, cos, -sin, zero, sin, cos)
    elif axis == 'Y':
        flat = (cos, zero, sin, zero, one, zero, -sin, zero, cos)
    else:
        flat = (cos, -sin, zero, sin, cos, zero, zero, zero, one)
    return torch.stack(flat, -1).reshape(angle.shape + (3, 3))


def euler_angles_to_matrix(euler_angles, convention):
    ms = [_axis_angle_rotation(c, e) for c, e in zip(convention, torch.unbind(euler_angles, -1))]
    return torch.matmul(torch.matmul(ms[0], ms[1]), ms[2])


def _index_from_letter(letter):
    return 'XYZ'.index(letter)


def _angle_from_tan(axis, other_axis, data, horizontal, tait_bryan):
    i1, i2 = {'X': (2, 1), 'Y': (0, 2), 'Z': (1, 0)}[axis]
    if horizontal:
        i2, i1 = i1, i2
    even = (axis + other_axis) in ['XY', 'YZ', 'ZX']
    if horizontal == even:
        return torch.atan2(data[..., i1], data[..., i2])
    if tait_bryan:
        return torch.atan2(-data[..., i2], data[..., i1])
    return torch.atan2(data[..., i2], -data[..., i1])


def matrix_to_euler_angles(matrix, convention):
    i0, i2 = _index_from_letter(convention[0]), _index_from_letter(convention[2])
    tait_bryan = i0 != i2
    if tait_bryan:
        central = torch.asin(matrix[..., i0, i2] * (-1.0 if i0 - i2 in [-1, 2] else 1.0))
    else:
        central = torch.acos(matrix[..., i0, i0])
    o = (_angle_from_tan(convention[0], convention[1], matrix[..., i2], False, tait_bryan), central,
         _angle_from_tan(convention[2], convention[1], matrix[..., i0, :], True, tait_bryan))
    return torch.stack(o, -1)


def _populate(module):
    n = module.__name__
    if n == 'pytorch3d.transforms':
        module.euler_angles_to_matrix = euler_angles_to_matrix
        module.matrix_to_euler_angles = matrix_to_euler_angles
    if n == 'mmengine.model':
        class BaseModule(torch.nn.Module):
            def __init__(self, init_cfg=None, *a, **k):
                super().__init__()
        module.BaseModule = BaseModule
        module.BaseModel = BaseModule
        module.bias_init_with_prob = lambda p: float(-torch.log(torch.tensor((1 - p) / p)))


    if n == 'mmengine.structures':
        class InstanceData:
            """attribute bag; len() = length of its first tensor-like field (what the assigner relies on)"""

            def __init__(self, **kw):
                self.__dict__.update(kw)

            def __len__(self):
                for v in self.__dict__.values():
                    if hasattr(v, '__len__'):
                        return len(v)
                return 0

            def __contains__(self, k):
                return k in self.__dict__
        module.InstanceData = InstanceData
    if n in ('mmengine', 'mmengine.fileio'):
        def load(path, *a, **k):
            """mmengine.load for the two formats the dataset class reads (.pkl, and .npy through numpy itself)"""
            import json
            import pickle
            if str(path).endswith('.json'):
                with open(path) as f:
                    return json.load(f)
            with open(path, 'rb') as f:
                return pickle.load(f)
        module.load = load
        module.track_iter_progress = lambda it, *a, **k: it
    if n == 'mmengine.dataset':
        import copy
        import os

        class BaseDataset:
            """the part of mmengine.dataset.BaseDataset (v0.10, base_dataset.py) the reference's dataset class relies
            on: metainfo copy, `_join_prefix` (ann_file and every data_prefix entry joined with data_root), eager
            `full_init` = load_data_list()"""

            def __init__(self, ann_file='', metainfo=None, data_root='', data_prefix=dict(img_path=''), pipeline=(),
                         test_mode=False, **kw):
                self.ann_file, self.data_root, self.test_mode = ann_file, data_root, test_mode
                self._metainfo = copy.deepcopy(dict(metainfo or {}))
                self.data_prefix = copy.copy(data_prefix)
                if self.ann_file and not os.path.isabs(self.ann_file) and self.data_root:
                    self.ann_file = os.path.join(self.data_root, self.ann_file)
                for k, v in list(self.data_prefix.items()):
                    if not os.path.isabs(v) and self.data_root:
                        self.data_prefix[k] = os.path.join(self.data_root, v)
                self.data_list = self.load_data_list()

            @property
            def metainfo(self):
                return self._metainfo

            def _serialize_data(self):
                return None, None
        module.BaseDataset = BaseDataset
    if n == 'mmcv.transforms':
        class BaseTransform:
            def __call__(self, results):
                return self.transform(results)

        class Compose:
            def __init__(self, transforms=()):
                self.transforms = list(transforms or ())

            def __call__(self, data):
                for t in self.transforms:
                    data = t(data)
                    if data is None:
                        return None
                return data
        module.BaseTransform = BaseTransform
        module.Compose = Compose
    if n == 'mmdet.models.task_modules':
        class AssignResult:
            def __init__(self, num_gts, gt_inds, max_overlaps, labels=None):
                self.num_gts, self.gt_inds, self.max_overlaps, self.labels = num_gts, gt_inds, max_overlaps, labels
        module.AssignResult = AssignResult
        module.BaseAssigner = object
    if n == 'mmdet.models.utils':
        from functools import partial

        def multi_apply(func, *args, **kwargs):
            pfunc = partial(func, **kwargs) if kwargs else func
            return tuple(map(list, zip(*map(pfunc, *args))))
        module.multi_apply = multi_apply
    if n == 'mmdet.utils':
        module.reduce_mean = lambda t: t
    if n == 'mmcv.cnn':
        module.Linear = torch.nn.Linear
    if n == 'pytorch3d.ops':
        def box3d_overlap(corners1, corners2, eps=1e-4):
            raise RuntimeError('bind pytorch3d.ops.box3d_overlap to the oracle restatement before use')
        module.box3d_overlap = box3d_overlap


def install(reference_root='/root/reference'):
    if not any(isinstance(f, _Finder) for f in sys.meta_path):
        sys.meta_path.insert(0, _Finder())
    if reference_root not in sys.path:
        sys.path.insert(0, reference_root)
