"""Golden vectors for the dataset reader, made by running the REFERENCE's own classes:
  * EmbodiedScanDataset (embodiedscan/datasets/embodiedscan_dataset.py) on the committed synthetic dataset
    tests/golden/fake_dataset/ (written by embodiedscan_amd.synth.write_dataset; regenerate with --write-dataset),
    in train mode, in test mode (eval_ann_info) and with remove_dontcare / a class subset;
  * MultiViewPipeline's view choice (transforms/multiview.py:46-64) for seeded np.random streams;
  * PointSample._points_random_sampling (transforms/points.py:155-206).
mmengine / mmcv are absent: oracle/_ref_stubs.py supplies BaseDataset (prefix joining + eager load), mmengine.load and
BaseTransform / Compose.  TEST INFRASTRUCTURE; run in the build container only (needs /root/reference):
    python oracle/make_golden_dataset.py [--write-dataset]"""
import os
import pickle
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
FIX = os.path.join(ROOT, 'tests', 'golden', 'fake_dataset')
FIX_KW = dict(n_scans=2, n_frames=5, height=60, width=80, n_boxes=6, seed=0, n_voxels=(8, 8, 4))


def plain(x):
    """reference objects -> plain python / numpy (boxes -> their (G, 9) tensor)"""
    import torch
    if hasattr(x, 'tensor') and torch.is_tensor(x.tensor):
        return x.tensor.numpy()
    if torch.is_tensor(x):
        return x.numpy()
    if isinstance(x, dict):
        return {k: plain(v) for k, v in x.items() if k != 'box_type_3d'}
    if isinstance(x, (list, tuple)):
        return [plain(v) for v in x]
    if isinstance(x, str):
        return x.replace(FIX, '<root>')
    return x


def main():
    if '--write-dataset' in sys.argv:
        import shutil
        from embodiedscan_amd import synth
        shutil.rmtree(FIX, ignore_errors=True)
        os.makedirs(FIX)
        synth.write_dataset(FIX, **FIX_KW)
    from oracle import _ref_stubs
    _ref_stubs.install()
    from embodiedscan.datasets.embodiedscan_dataset import EmbodiedScanDataset as Ref
    from embodiedscan.datasets.transforms.multiview import MultiViewPipeline
    from embodiedscan.datasets.transforms.points import PointSample
    from embodiedscan_amd.synth import _NOUNS
    names = _NOUNS + ['object']
    out = {}
    cases = dict(train=dict(metainfo=dict(classes=names)),
                 test=dict(metainfo=dict(classes=names), test_mode=True),
                 subset_nodontcare=dict(metainfo=dict(classes=names[:5], occ_classes=names[:7]), remove_dontcare=True),
                 default_classes=dict(metainfo=dict(occ_classes=names)))
    for tag, kw in cases.items():
        ds = Ref(data_root=FIX, ann_file='embodiedscan_infos_train.pkl', pipeline=[], **kw)
        out[tag] = dict(data_list=plain(ds.data_list), label_mapping=ds.label_mapping, occ_label_mapping=ds.occ_label_mapping,
                        classes=list(ds.metainfo['classes']))
    # visual grounding reader
    from embodiedscan.datasets.mv_3dvg_dataset import MultiView3DGroundingDataset as RefVG
    for tag, kw in dict(vg_train=dict(metainfo=dict(classes=names), tokens_positive_rebuild=True),
                        vg_test=dict(metainfo=dict(classes=names), test_mode=True, tokens_positive_rebuild=False)).items():
        ds = RefVG(data_root=FIX, ann_file='embodiedscan_infos_train.pkl', vg_file='embodiedscan_train_vg.json', pipeline=[], **kw)
        out[tag] = dict(data_list=plain(ds.data_list), label_mapping=ds.label_mapping)
    # view choice: extrinsic i is tagged with its index so the chosen ids can be read back
    views = []
    for seed, n_total, n_images, ordered in ((0, 30, 20, False), (1, 12, 20, False), (2, 300, 50, True), (3, 40, 50, True),
                                             (4, 50, 50, True), (5, 5, 3, False)):
        mv = MultiViewPipeline(transforms=[], n_images=n_images, ordered=ordered)
        res = dict(img_path=[f'{i}.jpg' for i in range(n_total)],
                   depth2img=dict(intrinsic=np.eye(4), extrinsic=[np.full((4, 4), i) for i in range(n_total)]))
        np.random.seed(seed)
        mv.transform(res)
        views.append(dict(seed=seed, n_total=n_total, n_images=n_images, ordered=ordered,
                          ids=np.array([int(e[0, 0]) for e in res['depth2img']['extrinsic']])))
    out['views'] = views
    samples = []
    for seed, n, num in ((0, 5000, 1000), (1, 300, 1000), (2, 1000, 1000), (3, 1, 10)):
        np.random.seed(seed)
        _, ch = PointSample(num_points=num)._points_random_sampling(np.arange(n), num, None, False, return_choices=True)
        samples.append(dict(seed=seed, n=n, num=num, choices=np.asarray(ch)))
    out['point_sample'] = samples
    with open(os.path.join(ROOT, 'tests', 'golden', 'dataset_parse.pkl'), 'wb') as f:
        pickle.dump(out, f, protocol=4)
    print('wrote dataset_parse.pkl:', {k: (len(v['data_list']) if isinstance(v, dict) else len(v)) for k, v in out.items()})


if __name__ == '__main__':
    main()
