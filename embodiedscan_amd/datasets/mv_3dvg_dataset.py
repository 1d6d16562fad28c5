"""Visual-grounding annotation reader (BASELINE config 4) behind the reference's registry name.

Mirrors `MultiView3DGroundingDataset` (embodiedscan/datasets/mv_3dvg_dataset.py:16-534): the scan infos are parsed like
the detection dataset's (without occupancy files; every instance key is carried under the reference's name mapping,
:484-523), then one sample per language annotation of the `vg_file` is assembled (`load_language_data`, :301-404):
prompt text, the target box(es) looked up by `bbox_id`, `tokens_positive` (optionally rebuilt from the target phrase),
the hard / unique / view-dependent flags.  Pinned to the reference's class by tests/golden/dataset_parse.pkl.
`load_scan()` decodes and draws like the detection reader and adds what `pipeline.make_grounding_batch` needs."""
import json
import os

import numpy as np

from ..registry import DATASETS
from .embodiedscan_dataset import EmbodiedScanDataset

_NAME_MAPPING = {'bbox_label_3d': 'gt_labels_3d', 'bbox_label': 'gt_bboxes_labels', 'bbox': 'gt_bboxes',
                 'bbox_3d': 'gt_bboxes_3d', 'depth': 'depths', 'center_2d': 'centers_2d', 'attr_label': 'attr_labels',
                 'velocity': 'velocities'}
_VIEW_DEP = ('front', 'behind', 'back', 'left', 'right', 'facing', 'leftmost', 'rightmost', 'looking', 'across')


@DATASETS.register_module()
class MultiView3DGroundingDataset(EmbodiedScanDataset):
    def __init__(self, data_root, ann_file, vg_file, metainfo=None, pipeline=(), box_type_3d='Euler-Depth',
                 serialize_data=False, filter_empty_gt=True, remove_dontcare=False, test_mode=False, load_eval_anns=True,
                 tokens_positive_rebuild=False, **kwargs):
        metainfo = dict(metainfo or {})
        self.tokens_positive_rebuild = tokens_positive_rebuild
        self._all_classes = metainfo.get('classes') == 'all'          # resolved against the info file's categories
        if self._all_classes:
            metainfo.pop('classes')
        super().__init__(data_root, ann_file, metainfo=metainfo, pipeline=pipeline, test_mode=test_mode,
                         load_eval_anns=load_eval_anns, filter_empty_gt=filter_empty_gt, remove_dontcare=remove_dontcare,
                         box_type_3d=box_type_3d, **kwargs)
        self.vg_file = os.path.join(self.data_root, vg_file)
        self.scans = {d['scan_id']: d for d in self.data_list}      # convert_info_to_scan (:236-240)
        self.data_list = self.load_language_data()

    def process_metainfo(self):
        if self._all_classes and 'classes' not in self._metainfo:
            # the reference substitutes its built-in 288-name tuple (METAINFO, :58-128); the shipped info files list the
            # same names as `categories`, which is what is available here
            self._metainfo['classes'] = list(self._metainfo['categories'].keys())
        super().process_metainfo()

    @staticmethod
    def _is_view_dep(text):
        words = set(text.split())
        return any(rel in words for rel in _VIEW_DEP)

    # ------------------------------------------------------------------ scan infos (mv_3dvg_dataset.py:406-534)
    def parse_data_info(self, info):
        info['box_type_3d'] = self.box_type_3d
        info['axis_align_matrix'] = self._get_axis_align_matrix(info)
        info['img_path'], info['depth_img_path'] = [], []
        info['scan_id'] = info['sample_idx']
        info['depth_shift'] = 4000.0 if info['sample_idx'].split('/')[0] == 'matterport3d' else 1000.0
        cam2img = info['cam2img'].astype(np.float32) if 'cam2img' in info else []
        extrinsics = []
        root = self.data_prefix.get('img_path', '')
        for im in info['images']:
            info['img_path'].append(os.path.join(root, im['img_path']))
            info['depth_img_path'].append(os.path.join(root, im['depth_path']))
            extrinsics.append(np.linalg.inv(info['axis_align_matrix'] @ im['cam2global']).astype(np.float32))
            if 'cam2img' not in info:
                cam2img.append(im['cam2img'].astype(np.float32))
        info['depth2img'] = dict(extrinsic=extrinsics, intrinsic=cam2img, origin=np.array([.0, .0, .5]).astype(np.float32))
        if 'depth_cam2img' not in info:
            info['depth_cam2img'] = cam2img
        if not self.test_mode:
            info['ann_info'] = self.parse_ann_info(info)
        if self.test_mode and self.load_eval_anns:
            info['ann_info'] = self.parse_ann_info(info)
            info['eval_ann_info'] = info['ann_info']
        return info

    def parse_ann_info(self, info):
        ann = None
        inst = info.get('instances') or []
        if len(inst) > 0:
            ann = {}
            for name in list(inst[0].keys()):
                vals = [it[name] for it in inst]
                if 'label' in name and name != 'attr_label':
                    vals = [self.label_mapping[v] for v in vals]
                if 'label' in name:
                    arr = np.array(vals).astype(np.int64)
                elif name in _NAME_MAPPING:
                    arr = np.array(vals).astype(np.float32)
                else:
                    arr = np.array(vals)
                ann[_NAME_MAPPING.get(name, name)] = arr
            ann['instances'] = info['instances']
        if ann is None:
            ann = dict(gt_bboxes_3d=np.zeros((0, 9), dtype=np.float32), gt_labels_3d=np.zeros((0,), dtype=np.int64))
        return ann

    # ------------------------------------------------------------------ language annotations (mv_3dvg_dataset.py:301-404)
    def load_language_data(self):
        with open(self.vg_file) as f:
            annotations = json.load(f)
        infos = []
        for anno in annotations:
            data = self.scans[anno['scan_id']]
            li = dict(scan_id=data['scan_id'], text=anno['text'], axis_align_matrix=data['axis_align_matrix'],
                      img_path=data['img_path'], depth_img_path=data['depth_img_path'], depth2img=data['depth2img'],
                      depth_shift=data['depth_shift'], depth_cam2img=data['depth_cam2img'])
            if 'cam2img' in data:
                li['cam2img'] = data['cam2img']
            ann_info = data['ann_info']
            la = dict(is_view_dep=self._is_view_dep(li['text']))
            labels, bboxes = ann_info['gt_labels_3d'], ann_info['gt_bboxes_3d']
            if 'target_id' in anno:
                li['target_id'] = anno['target_id']
                object_ids = ann_info['bbox_id']
                if isinstance(anno['target_id'], int):
                    ind = np.where(object_ids == li['target_id'])[0]
                    if len(ind) != 1:
                        continue                                   # target missing (or ambiguous) in this scan: dropped
                    la['gt_bboxes_3d'], la['gt_labels_3d'] = bboxes[ind], labels[ind]
                    if 'tokens_positive' in anno:
                        if self.tokens_positive_rebuild:
                            anno['tokens_positive'] = [[anno['text'].find(part), anno['text'].find(part) + len(part)]
                                                       for part in anno['target'].split()]
                        li['tokens_positive'] = [anno['tokens_positive']]
                elif isinstance(anno['target_id'], list):
                    inds, keep, unique = [], [], True
                    for j, tid in enumerate(li['target_id']):
                        ind = np.where(object_ids == tid)[0]
                        if len(ind) != 1:
                            unique = False
                            break
                        keep.append(j)
                        inds.append(ind[0])
                    if not unique:
                        continue
                    la['gt_bboxes_3d'], la['gt_labels_3d'] = bboxes[inds], labels[inds]
                    if 'tokens_positive' in anno:
                        li['tokens_positive'] = [[anno['tokens_positive'][j]] for j in keep]
                else:
                    raise NotImplementedError
                if 'distractor_ids' in anno:
                    li['distractor_ids'] = anno['distractor_ids']
                # (as the reference: an annotation with a target but no distractor list is a KeyError, :385-389)
                la['is_hard'] = len(li['distractor_ids']) > 3
                la['is_unique'] = len(li['distractor_ids']) == 0
            else:
                la['gt_bboxes_3d'], la['gt_labels_3d'] = bboxes, labels
                la['is_hard'] = la['is_unique'] = False
            if not self.test_mode:
                li['ann_info'] = la
            if self.test_mode and self.load_eval_anns:
                li['ann_info'] = la
                li['eval_ann_info'] = li['ann_info']
            infos.append(li)
        del self.scans
        return infos

    def load_scan(self, idx, rng=None, alloc=None):
        info = dict(self.data_list[idx])
        info.setdefault('sample_idx', info['scan_id'])
        scan = self.pipeline(info, rng if rng is not None else np.random, alloc)
        scan['text'] = info['text']
        if 'tokens_positive' in info:
            scan['tokens_positive'] = info['tokens_positive']
        ann = info.get('ann_info') or {}
        for k in ('is_view_dep', 'is_hard', 'is_unique'):
            if k in ann:
                scan['meta'][k] = ann[k]
        return scan
