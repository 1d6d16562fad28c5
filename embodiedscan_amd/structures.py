"""The few data containers the model boundary needs (stand-ins for mmengine.structures.InstanceData,
Det3DDataSample and embodiedscan.structures.EulerDepthInstance3DBoxes: only what the train step reads:
`.tensor`, `.gravity_center`, `.volume`, `.with_yaw`; euler_box3d.py:24-58,137-140, base_box3d.py:87-90)."""
import torch


class EulerDepthInstance3DBoxes:
    with_yaw = True

    def __init__(self, tensor, box_dim=9, origin=(0.5, 0.5, 0.5)):
        tensor = torch.as_tensor(tensor, dtype=torch.float32)
        if tensor.numel() == 0:
            tensor = tensor.reshape((0, box_dim))
        if tensor.shape[-1] == 6:
            tensor = torch.cat((tensor, tensor.new_zeros(tensor.shape[0], 3)), -1)
        elif tensor.shape[-1] == 7:
            tensor = torch.cat((tensor, tensor.new_zeros(tensor.shape[0], 2)), -1)
        assert tensor.shape[-1] == 9
        self.tensor = tensor.clone()
        if origin != (0.5, 0.5, 0.5):
            self.tensor[:, :3] += self.tensor[:, 3:6] * (self.tensor.new_tensor((0.5, 0.5, 0.5)) - self.tensor.new_tensor(origin))

    @property
    def gravity_center(self):
        return self.tensor[:, :3]

    @property
    def volume(self):
        return self.tensor[:, 3] * self.tensor[:, 4] * self.tensor[:, 5]

    def to(self, device):
        b = EulerDepthInstance3DBoxes(self.tensor.to(device))
        return b

    def __len__(self):
        return self.tensor.shape[0]

    @property
    def shape(self):
        return self.tensor.shape


class InstanceData:
    def __init__(self, **kw):
        self.__dict__.update(kw)

    def __contains__(self, k):
        return k in self.__dict__


class Det3DDataSample:
    def __init__(self, metainfo=None, gt_instances_3d=None):
        self.metainfo = dict(metainfo or {})
        self.gt_instances_3d = gt_instances_3d or InstanceData()

    def set_metainfo(self, d):
        self.metainfo.update(d)

    def get(self, k, default=None):
        return getattr(self, k, default)
