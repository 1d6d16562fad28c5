"""Annotation reader for the EmbodiedScan `.pkl` info files, behind the reference's registry name.

Mirrors `EmbodiedScanDataset` (embodiedscan/datasets/embodiedscan_dataset.py:14-375): same constructor arguments, same
`process_metainfo` / `parse_data_info` / `parse_ann_info` results (key for key; pinned to the reference's own class by
tests/golden/dataset_parse.pkl, made by oracle/make_golden_dataset.py).  What differs is what happens after parsing: the
reference hands each info dict to a CPU transform pipeline; here `load_scan()` only DECODES the files and draws the
pipeline's random decisions (host, datasets/loading.py) and returns the raw scan the device path consumes (A1-A3 +
augmentation run on the GPU, pipeline.py / csrc/data.hip).  No mmengine BaseDataset behind it: lazy init, serialisation
for fork workers and the `RepeatDataset` wrapper are control plane (out of scope)."""
import os
import pickle
import warnings

import numpy as np

from ..registry import DATASETS
from . import loading


@DATASETS.register_module()
class EmbodiedScanDataset:
    def __init__(self, data_root, ann_file, metainfo=None, pipeline=(), test_mode=False, load_eval_anns=True,
                 filter_empty_gt=True, remove_dontcare=False, box_type_3d='Euler-Depth', data_prefix=None, **kwargs):
        if box_type_3d.lower() != 'euler-depth':
            raise NotImplementedError(f'box_type_3d={box_type_3d!r}: only the 9-DoF Euler-Depth boxes of the shipped configs')
        self.box_type_3d = box_type_3d
        self.data_root = data_root
        self.ann_file = ann_file if os.path.isabs(ann_file) or not data_root else os.path.join(data_root, ann_file)
        # mmengine joins every data_prefix entry with data_root (BaseDataset._join_prefix); default img_path=''
        prefix = dict(img_path='') if data_prefix is None else dict(data_prefix)
        self.data_prefix = {k: (os.path.join(data_root, v) if data_root and not os.path.isabs(v) else v)
                            for k, v in prefix.items()}
        self.test_mode = test_mode
        self.load_eval_anns = load_eval_anns
        self.filter_empty_gt = filter_empty_gt
        self.remove_dontcare = remove_dontcare
        self._metainfo = dict(metainfo or {})
        self.pipeline = loading.ScanPipeline.from_cfg(pipeline)
        self.data_list = self.load_data_list()

    @property
    def metainfo(self):
        return self._metainfo

    # ------------------------------------------------------------------ metainfo (embodiedscan_dataset.py:62-85)
    def process_metainfo(self):
        assert 'categories' in self._metainfo
        cats = self._metainfo['categories']
        if 'classes' not in self._metainfo:
            self._metainfo.setdefault('classes', list(cats.keys()))
        classes = list(self._metainfo['classes'])
        self.label_mapping = np.full(max(cats.values()) + 1, -1, dtype=int)
        for key, value in cats.items():
            if key in classes:
                self.label_mapping[value] = classes.index(key)
        self.occ_label_mapping = np.full(max(cats.values()) + 1, -1, dtype=int)
        for idx, name in enumerate(self._metainfo.get('occ_classes', ())):
            self.occ_label_mapping[cats[name]] = idx + 1          # 1-based, 0 is empty

    # ------------------------------------------------------------------ one info dict (embodiedscan_dataset.py:87-156)
    def parse_data_info(self, info):
        info['box_type_3d'] = self.box_type_3d
        info['axis_align_matrix'] = self._get_axis_align_matrix(info)
        info['scan_id'] = info['sample_idx']
        ann_dataset = info['sample_idx'].split('/')[0]
        info['depth_shift'] = 4000.0 if ann_dataset == 'matterport3d' else 1000.0
        info['img_path'], info['depth_img_path'] = [], []
        cam2img = info['cam2img'].astype(np.float32) if 'cam2img' in info else []
        extrinsics = []
        root = self.data_prefix.get('img_path', '')
        for im in info['images']:
            info['img_path'].append(os.path.join(root, im['img_path']))
            info['depth_img_path'].append(os.path.join(root, im['depth_path']))
            align_global2cam = np.linalg.inv(info['axis_align_matrix'] @ im['cam2global'])
            extrinsics.append(align_global2cam.astype(np.float32))
            if 'cam2img' not in info:
                cam2img.append(im['cam2img'].astype(np.float32))
        info['depth2img'] = dict(extrinsic=extrinsics, intrinsic=cam2img, origin=np.array([.0, .0, .5]).astype(np.float32))
        if 'depth_cam2img' not in info:
            info['depth_cam2img'] = cam2img
        if not self.test_mode:
            info['ann_info'] = self.parse_ann_info(info)
            if self.filter_empty_gt and 'gt_occupancy' in info['ann_info'] \
                    and info['ann_info']['gt_occupancy'].shape[0] == 0:
                return None                                        # scans without occupancy ground truth are dropped
        if self.test_mode and self.load_eval_anns:
            info['ann_info'] = self.parse_ann_info(info)
            info['eval_ann_info'] = self._remove_dontcare(info['ann_info'])
        return info

    def _occupancy_files(self, sample_idx):
        """(occupancy.npy, visible_occupancy.pkl) of a scan (embodiedscan_dataset.py:201-230)"""
        root = self.data_prefix.get('img_path', '')
        parts = sample_idx.split('/')
        ds = parts[0]
        if ds == 'scannet':
            d = os.path.join(root, ds, 'scans', parts[1], 'occupancy')
            return os.path.join(d, 'occupancy.npy'), os.path.join(d, 'visible_occupancy.pkl')
        if ds == '3rscan':
            d = os.path.join(root, ds, parts[1], 'occupancy')
            return os.path.join(d, 'occupancy.npy'), os.path.join(d, 'visible_occupancy.pkl')
        if ds == 'matterport3d':
            d = os.path.join(root, ds, parts[1], 'occupancy')
            return os.path.join(d, f'occupancy_{parts[2]}.npy'), os.path.join(d, f'visible_occupancy_{parts[2]}.pkl')
        if ds == 'arkitscenes':
            return None, None
        raise NotImplementedError(ds)

    # ------------------------------------------------------------------ annotations (embodiedscan_dataset.py:158-253)
    def parse_ann_info(self, info):
        inst = info.get('instances') or []
        ann = dict(gt_bboxes_3d=np.zeros((len(inst), 9), dtype=np.float32), gt_labels_3d=np.zeros((len(inst),), dtype=np.int64))
        for i, ins in enumerate(inst):
            ann['gt_bboxes_3d'][i] = ins['bbox_3d']
            ann['gt_labels_3d'][i] = self.label_mapping[ins['bbox_label_3d']]
        if 'visible_instance_ids' in info['images'][0]:
            ids = [im['visible_instance_ids'] for im in info['images']]
            ann['visible_instance_masks'] = self._ids2masks(ids, ann['gt_labels_3d'].shape[0])
        if self.remove_dontcare:
            ann = self._remove_dontcare(ann)
        occ_file, mask_file = self._occupancy_files(info['sample_idx'])
        if occ_file is None:
            gt_occ = np.zeros((0, 4), dtype=np.int64)
        else:
            gt_occ = np.load(occ_file)
            cls = self.occ_label_mapping[gt_occ[:, 3]]            # categories not in occ_classes -> 255 (ignored)
            gt_occ[:, 3] = np.where(cls < 0, 255, cls)
        ann['gt_occupancy'] = gt_occ
        if mask_file is None:
            ann['visible_occupancy_masks'] = [[] for _ in info['images']]
        else:
            with open(mask_file, 'rb') as f:
                occ_masks = pickle.load(f)
            ann['visible_occupancy_masks'] = [occ_masks[i]['visible_occupancy'] for i in range(len(info['images']))]
        # the reference wraps gt_bboxes_3d in EulerDepthInstance3DBoxes(origin=(.5,.5,.5)): centre == the stored centre, so the
        # (G, 9) array is kept as is and wrapped when the data sample is built (pipeline.make_batch)
        return ann

    @staticmethod
    def _get_axis_align_matrix(info):
        if 'axis_align_matrix' in info:
            return np.array(info['axis_align_matrix'])
        warnings.warn('axis_align_matrix is not found in ScanNet data info, please use new pre-process scripts to '
                      're-generate ScanNet data')
        return np.eye(4).astype(np.float32)

    @staticmethod
    def _ids2masks(ids, mask_length):
        masks = []
        for v in ids:
            m = np.zeros((mask_length,), dtype=bool)
            m[v] = 1
            masks.append(m)
        return masks

    @staticmethod
    def _remove_dontcare(ann):
        """drop the instances whose label is -1 (embodiedscan_dataset.py:285-313); occupancy keys are not copied"""
        out, keep = {}, ann['gt_labels_3d'] > -1
        for key, v in ann.items():
            if key == 'instances':
                out[key] = v
            elif key == 'visible_instance_masks':
                out[key] = [m[keep] for m in v]
            elif key in ('gt_occupancy', 'visible_occupancy_masks'):
                pass
            else:
                out[key] = v[keep]
        return out

    # ------------------------------------------------------------------ the info file (embodiedscan_dataset.py:315-375)
    def load_data_list(self):
        with open(self.ann_file, 'rb') as f:
            annotations = pickle.load(f)
        if not isinstance(annotations, dict):
            raise TypeError(f'The annotations loaded from annotation file should be a dict, but got {type(annotations)}!')
        if 'data_list' not in annotations or 'metainfo' not in annotations:
            raise ValueError('Annotation must have data_list and metainfo keys')
        for k, v in annotations['metainfo'].items():
            self._metainfo.setdefault(k, v)
        self.process_metainfo()
        data_list = []
        for raw in annotations['data_list']:
            info = self.parse_data_info(raw)
            if isinstance(info, dict):
                data_list.append(info)
            elif info is not None:
                raise TypeError(f'data_info should be a dict or None, but got {type(info)}')
        return data_list

    def __len__(self):
        return len(self.data_list)

    def get_data_info(self, idx):
        return self.data_list[idx]

    # ------------------------------------------------------------------ hand-over to the device path
    def load_scan(self, idx, rng=None, alloc=None):
        """decode scan `idx` and draw the pipeline's random decisions -> the raw scan dict of pipeline.pin_scan /
        upload_scan (see loading.ScanPipeline).  rng: numpy RandomState (the reference uses the global np.random stream
        in the same order of draws).  alloc: where the frames are decoded into (ScanPipeline.__call__)."""
        return self.pipeline(self.data_list[idx], rng if rng is not None else np.random, alloc)

    def __getitem__(self, idx):
        return self.load_scan(idx)
