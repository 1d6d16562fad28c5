"""Seeded synthetic RGB-D scans of the shape the mv-3ddet pipeline consumes.

There is no dataset in this environment, so bench.py, smoke() and the parity tests
all draw from this generator: an axis-aligned room with rotated furniture boxes,
V posed pinhole cameras, analytic ray-box z-depth, uint8 noise for RGB, the
furniture as 9-DoF ground truth, and the host-side random decisions of the
reference data pipeline (PointSample indices, RandomFlip3D, GlobalRotScaleTrans:
configs/detection/mv-det3d_...py:134-160) drawn here so that the device path and
the CPU oracle consume identical inputs.
"""
import math
import numpy as np


def _euler_zxy(a):
    ca, sa, cb, sb, cc, sc = math.cos(a[0]), math.sin(a[0]), math.cos(a[1]), math.sin(a[1]), math.cos(a[2]), math.sin(a[2])
    rz = np.array([[ca, -sa, 0], [sa, ca, 0], [0, 0, 1.]])
    rx = np.array([[1., 0, 0], [0, cb, -sb], [0, sb, cb]])
    ry = np.array([[cc, 0, sc], [0, 1., 0], [-sc, 0, cc]])
    return rz @ rx @ ry


def _mat_to_euler_zxy(m):
    return np.array([math.atan2(-m[0, 1], m[1, 1]), math.asin(max(-1., min(1., m[2, 1]))), math.atan2(-m[2, 0], m[2, 2])])


def _ray_boxes(o, d, c, half, R):
    """Slab test of P rays against nb oriented boxes at once (torch, any device).
    o (3,), d (P,3), c (nb,3), half (nb,3), R (nb,3,3) -> nearest hit distance t (P,) (inf = miss)."""
    import torch
    oo = torch.einsum('bj,bjk->bk', o[None] - c, R)            # R^T (o - c)            (nb,3)
    dd = torch.einsum('pj,bjk->pbk', d, R)                       # (P,nb,3)
    dd = torch.where(dd.abs() < 1e-12, torch.full_like(dd, 1e-12), dd)
    t1 = (-half[None] - oo[None]) / dd
    t2 = (half[None] - oo[None]) / dd
    tmin = torch.minimum(t1, t2).amax(-1)
    tmax = torch.maximum(t1, t2).amin(-1)
    hit = (tmax >= tmin.clamp(min=0)) & (tmax > 0)
    t = torch.where(tmin > 0, tmin, tmax)                        # inside the box -> exit distance
    t = torch.where(hit, t, torch.full_like(t, float('inf')))
    return t.amin(1)


def make_scan(seed, n_views=20, height=480, width=640, img_size=(480, 480), n_boxes=25, n_points=100000,
              n_classes=284, augment=True, max_depth=6.0, render_device='cpu'):
    """One synthetic scan.  Returns a dict of numpy arrays + the meta dict the
    reference attaches to a data sample (depth2img, scale_factor, img_shape, pcd_* ...)."""
    rng = np.random.default_rng(seed)
    room_c, room_h = np.array([0., 0., 1.4]), np.array([3., 2.5, 1.4])
    sizes = rng.uniform(0.3, 1.6, (n_boxes, 3))
    centers = np.stack([rng.uniform(-2.6, 2.6, n_boxes), rng.uniform(-2.1, 2.1, n_boxes), sizes[:, 2] / 2 + rng.uniform(0, 0.8, n_boxes)], 1)
    eulers = np.stack([rng.uniform(-math.pi, math.pi, n_boxes), rng.uniform(-.1, .1, n_boxes), rng.uniform(-.1, .1, n_boxes)], 1)
    labels = rng.integers(0, n_classes, n_boxes)
    rots = [_euler_zxy(e) for e in eulers]

    fx = fy = 577.87 * width / 640.0
    cx, cy = (width - 1) / 2.0, (height - 1) / 2.0
    K = np.array([[fx, 0, cx, 0], [0, fy, cy, 0], [0, 0, 1, 0], [0, 0, 0, 1.]])
    us, vs = np.meshgrid(np.arange(width), np.arange(height))
    dirs_cam = np.stack([(us.ravel() - cx) / fx, (vs.ravel() - cy) / fy, np.ones(width * height)], 1)

    import torch
    depth = np.zeros((n_views, height, width), np.float32)
    extr, intr = [], []
    rd = torch.device(render_device)
    all_c = torch.tensor(np.concatenate([room_c[None], centers]), dtype=torch.float64, device=rd)
    all_h = torch.tensor(np.concatenate([room_h[None], sizes / 2]), dtype=torch.float64, device=rd)
    all_R = torch.tensor(np.stack([np.eye(3)] + rots), dtype=torch.float64, device=rd)
    dirs_t = torch.tensor(dirs_cam, dtype=torch.float64, device=rd)
    for v in range(n_views):
        pos = np.array([rng.uniform(-2.2, 2.2), rng.uniform(-1.8, 1.8), rng.uniform(1.0, 1.8)])
        yaw, pitch = rng.uniform(0, 2 * math.pi), rng.uniform(-0.5, 0.1)
        fwd = np.array([math.cos(yaw) * math.cos(pitch), math.sin(yaw) * math.cos(pitch), math.sin(pitch)])
        right = np.cross(fwd, np.array([0, 0, 1.]))
        right /= np.linalg.norm(right)
        down = np.cross(fwd, right)
        c2w = np.eye(4)
        c2w[:3, 0], c2w[:3, 1], c2w[:3, 2], c2w[:3, 3] = right, down, fwd, pos
        w2c = np.linalg.inv(c2w)
        d = dirs_t @ torch.tensor(c2w[:3, :3].T, dtype=torch.float64, device=rd)
        pos_t = torch.tensor(pos, dtype=torch.float64, device=rd)
        chunks = []
        step = 65536 if rd.type == 'cpu' else d.shape[0]
        for c0 in range(0, d.shape[0], step):
            chunks.append(_ray_boxes(pos_t, d[c0:c0 + step], all_c, all_h, all_R))
        t = torch.cat(chunks).cpu().numpy()
        z = np.where(np.isfinite(t), t, 0.)          # camera-frame z == t because dirs_cam.z == 1
        z = np.where(z > max_depth, 0., z)
        depth[v] = z.reshape(height, width).astype(np.float32)
        extr.append(w2c.astype(np.float32))
        intr.append(K.astype(np.float32))

    # PointSample(n_points // 10) per view on the non-zero pixels, then PointSample(n_points)
    per_view = n_points // n_views * 2
    sel_view, sel_pix = [], []
    for v in range(n_views):
        nz = np.nonzero(depth[v].reshape(-1))[0]
        if len(nz) == 0:
            nz = np.array([0])
        pick = rng.choice(len(nz), per_view, replace=len(nz) < per_view)
        sel_view.append(np.full(per_view, v, np.int32))
        sel_pix.append(nz[pick].astype(np.int32))
    sel_view, sel_pix = np.concatenate(sel_view), np.concatenate(sel_pix)
    pick2 = rng.choice(len(sel_pix), n_points, replace=len(sel_pix) < n_points)
    sel_view, sel_pix = sel_view[pick2], sel_pix[pick2]

    meta = dict(depth2img=dict(extrinsic=extr, intrinsic=intr, origin=np.zeros(3, np.float32)),
                img_shape=(img_size[0], img_size[1]), scale_factor=(img_size[1] / width, img_size[0] / height),
                box_type_3d='euler-depth', transformation_3d_flow=[])
    gt = np.concatenate([centers, sizes, eulers], 1)
    aug = dict(hflip=False, vflip=False, rot=np.eye(3, dtype=np.float32), scale=1.0, trans=np.zeros(3, np.float32))
    if augment:
        # RandomFlip3D(flip_ratio 0.5/0.5) then GlobalRotScaleTrans (augmentation.py:87-139,322-348)
        hf, vf = bool(rng.random() < .5), bool(rng.random() < .5)
        ang = -rng.uniform(-0.087266, 0.087266)
        scale = float(rng.uniform(.9, 1.1))
        trans = rng.normal(scale=.1, size=3).astype(np.float32)
        rz = _euler_zxy([ang, 0, 0])
        rot_mat_T = rz.T.astype(np.float32)             # points @ rot_mat_T  (base_points.py:198-201)
        M = np.diag([-1. if hf else 1., -1. if vf else 1., 1.])
        A = rz @ M                                      # p_aug = scale * A p + trans
        new = []
        for b in range(n_boxes):
            c = scale * (A @ centers[b]) + trans
            Rb = A @ rots[b] @ M                        # keep a proper rotation (mirror the box frame too)
            new.append(np.concatenate([c, sizes[b] * scale, _mat_to_euler_zxy(Rb)]))
        gt = np.stack(new)
        meta.update(pcd_horizontal_flip=hf, pcd_vertical_flip=vf, pcd_rotation=rot_mat_T, pcd_scale_factor=scale,
                    pcd_trans=trans)
        flow = (['HF'] if hf else []) + (['VF'] if vf else []) + ['R', 'S', 'T']
        meta['transformation_3d_flow'] = flow
        aug = dict(hflip=hf, vflip=vf, rot=rot_mat_T, scale=scale, trans=trans)
    img = rng.integers(0, 256, (n_views, 3, img_size[0], img_size[1]), dtype=np.uint8)
    return dict(depth=depth, img=img, extrinsic=np.stack(extr), intrinsic=np.stack(intr), sel_view=sel_view,
                sel_pix=sel_pix, gt_boxes=gt.astype(np.float32), gt_labels=labels.astype(np.int64), meta=meta, aug=aug)


def make_occ_gt(scan, n_voxels=(40, 40, 16), prior_range=(-3.2, -3.2, -1.28, 3.2, 3.2, 1.28), n_classes=81, seed=0,
                visible_frac=0.85):
    """Synthetic occupancy ground truth for BASELINE config 5: the scan's furniture voxelised on the
    (X,Y,Z) grid, `gt_occupancy` (N,4) int64 rows {x, y, z, label in 1..n_classes-1} (what LoadAnnotations3D hands to
    ImVoxelOccHead.loss, datasets/transforms/loading.py:428) and a boolean visibility mask `gt_occupancy_masks` (X,Y,Z)
    (ConstructMultiViewMasks, datasets/transforms/multiview.py:250-273)."""
    rng = np.random.default_rng(seed)
    X, Y, Z = n_voxels
    lo, hi = np.asarray(prior_range[:3]), np.asarray(prior_range[3:])
    cell = (hi - lo) / np.asarray(n_voxels)
    ix, iy, iz = np.meshgrid(np.arange(X), np.arange(Y), np.arange(Z), indexing='ij')
    idx = np.stack([ix.ravel(), iy.ravel(), iz.ravel()], 1)
    centres = lo[None] + (idx + 0.5) * cell[None]
    origin = np.asarray(scan['meta']['depth2img'].get('origin', np.zeros(3)), np.float64)
    centres = centres + origin[None]
    label = np.zeros(len(idx), np.int64)
    cls = rng.integers(1, n_classes, len(scan['gt_boxes']))
    for b, c in zip(scan['gt_boxes'].astype(np.float64), cls):
        R = _euler_zxy(b[6:9])
        local = (centres - b[None, :3]) @ R                      # R^T (p - c)
        inside = (np.abs(local) <= b[None, 3:6] / 2 + cell[None] * 0.25).all(1)
        label[inside] = c
    occ = np.concatenate([idx[label > 0], label[label > 0, None]], 1)
    occ = occ[rng.permutation(len(occ))]
    dup = occ[rng.integers(0, max(len(occ), 1), min(len(occ), 16))] if len(occ) else occ     # duplicate rows: last wins
    if len(dup):
        dup = dup.copy()
        dup[:, 3] = rng.integers(1, n_classes, len(dup))
        occ = np.concatenate([occ, dup], 0)
    mask = rng.random((X, Y, Z)) < visible_frac
    return dict(gt_occupancy=occ.astype(np.int64), gt_occupancy_masks=mask)


_NOUNS = ['chair', 'table', 'cabinet', 'sofa', 'lamp', 'shelf', 'desk', 'bed', 'monitor', 'plant', 'stool', 'box']
_ADJ = ['red', 'wooden', 'small', 'tall', 'dark', 'white', 'round', 'old']


def make_grounding_sample(scan, seed=0, max_targets=3):
    """Synthetic language annotation for BASELINE config 4: a prompt of 8-40 words naming 1..max_targets of
    the scan's boxes, each with one positive character span.  Returns dict(text, tokens_positive [[(beg, end)]] per
    target, gt_boxes (G,9) f32, gt_labels (G,) int64) -- the fields MultiView3DGroundingDataset attaches to a data sample
    (`text`, `tokens_positive`, `gt_instances_3d`)."""
    rng = np.random.default_rng(seed)
    n = len(scan['gt_boxes'])
    G = int(rng.integers(1, min(max_targets, n) + 1))
    pick = rng.choice(n, G, replace=False)
    words, spans = ['find'], []
    for k in range(G):
        phrase = f'{_ADJ[int(rng.integers(len(_ADJ)))]} {_NOUNS[int(rng.integers(len(_NOUNS)))]}'
        words.append('the')
        beg = len(' '.join(words)) + 1
        words.append(phrase)
        spans.append([(beg, beg + len(phrase))])
        words.append('and' if k + 1 < G else 'in')
    filler = ['the', 'room', 'that', 'is', 'close', 'to', 'the', 'wall', 'next', 'to', 'a', 'window', 'on', 'the', 'left', 'side',
              'of', 'the', 'door']
    words += filler[:int(rng.integers(2, len(filler)))]
    return dict(text=' '.join(words), tokens_positive=spans, gt_boxes=scan['gt_boxes'][pick].astype(np.float32),
                gt_labels=np.zeros(G, np.int64))


def write_dataset(root, n_scans=2, n_frames=6, height=60, width=80, n_boxes=6, class_names=None, seed=0,
                  ann_name='embodiedscan_infos_train.pkl', occupancy=True, n_voxels=(40, 40, 16), jpeg_quality=90,
                  render_device='cpu', depth_div=1):
    """Write a small synthetic dataset in the EmbodiedScan on-disk layout: the info `.pkl` the reference's
    EmbodiedScanDataset reads (embodiedscan_dataset.py:315-375: `metainfo.categories`, `data_list[*]` with
    sample_idx / axis_align_matrix / cam2img / depth_cam2img / images[*]{img_path, depth_path, cam2global,
    visible_instance_ids} / instances[*]{bbox_3d, bbox_label_3d, bbox_id}), JPEG colour frames, 16-bit PNG depth in
    millimetres (depth_shift 1000, :103-107), and per-scan occupancy.npy / visible_occupancy.pkl (:201-244).
    Geometry comes from make_scan (unaugmented); returns the list of source scans for round-trip checks.
    depth_div > 1: the depth PNGs are written at 1/depth_div of the colour resolution with their own `depth_cam2img`
    (real scans: ScanNet colour 1296x968, depth 640x480)."""
    import os
    import pickle
    from PIL import Image
    rng = np.random.default_rng(seed)
    class_names = list(class_names or (_NOUNS + ['object']))
    categories = {n: 3 * i + 1 for i, n in enumerate(class_names)}            # sparse ids: exercises label_mapping
    categories['unlabelled thing'] = 3 * len(class_names) + 5                 # a category outside `classes` -> label -1
    data_list, scans = [], []
    for s in range(n_scans):
        scan = make_scan(seed * 1000 + s, n_views=n_frames, height=height, width=width, img_size=(height, width),
                         n_boxes=n_boxes, n_points=2000, n_classes=len(class_names), augment=False,
                         render_device=render_device)
        scans.append(scan)
        name = f'scene{s:04d}_00'
        sample_idx = f'scannet/{name}'
        ang = rng.uniform(-math.pi, math.pi)
        A = np.eye(4)
        A[:3, :3] = _euler_zxy([ang, 0, 0])
        A[:3, 3] = rng.uniform(-1, 1, 3)
        Ainv = np.linalg.inv(A)
        frame_dir = os.path.join(root, 'scannet', 'posed_images', name)
        os.makedirs(frame_dir, exist_ok=True)
        images = []
        yy, xx = np.meshgrid(np.arange(height), np.arange(width), indexing='ij')
        for v in range(n_frames):
            base = np.stack([(xx * (3 + v) + s * 17) % 256, (yy * (2 + v)) % 256, ((xx + yy) * 2 + 40 * v) % 256], -1)
            img = np.clip(base + rng.integers(-12, 13, base.shape), 0, 255).astype(np.uint8)
            Image.fromarray(img).save(os.path.join(frame_dir, f'{v:05d}.jpg'), quality=jpeg_quality)
            mm = np.clip(np.rint(scan['depth'][v] * 1000.0), 0, 65535).astype(np.uint16)[::depth_div, ::depth_div]
            Image.fromarray(np.ascontiguousarray(mm)).save(os.path.join(frame_dir, f'{v:05d}.png'))
            c2w = np.linalg.inv(scan['extrinsic'][v].astype(np.float64))
            vis = sorted(rng.choice(n_boxes, rng.integers(1, n_boxes + 1), replace=False).tolist())
            images.append(dict(img_path=f'scannet/posed_images/{name}/{v:05d}.jpg',
                               depth_path=f'scannet/posed_images/{name}/{v:05d}.png',
                               cam2global=Ainv @ c2w, visible_instance_ids=vis))
        instances = []
        for b in range(n_boxes):
            lab = categories['unlabelled thing'] if (b == n_boxes - 1 and s == 0) else categories[class_names[int(scan['gt_labels'][b])]]
            instances.append(dict(bbox_3d=scan['gt_boxes'][b].astype(np.float64).tolist(), bbox_label_3d=int(lab), bbox_id=b + 1))
        dK = scan['intrinsic'][0].astype(np.float64).copy()
        dK[:2, :3] /= depth_div                 # pixel (u, v) of the decimated map is pixel (u, v) * depth_div of the full one
        data_list.append(dict(sample_idx=sample_idx, axis_align_matrix=A, cam2img=scan['intrinsic'][0].astype(np.float64),
                              depth_cam2img=dK, images=images, instances=instances))
        if occupancy:
            occ = make_occ_gt(scan, n_voxels=n_voxels, n_classes=len(class_names) + 1, seed=seed + s)
            g = occ['gt_occupancy'].copy()
            ids = np.array([categories[n] for n in class_names])
            g[:, 3] = ids[g[:, 3] - 1]                                        # file stores category ids
            g[: min(3, len(g)), 3] = categories['unlabelled thing']           # -> 255 (ignored) after mapping
            occ_dir = os.path.join(root, 'scannet', 'scans', name, 'occupancy')
            os.makedirs(occ_dir, exist_ok=True)
            np.save(os.path.join(occ_dir, 'occupancy.npy'), g)
            masks = [dict(visible_occupancy=(rng.random(n_voxels) < 0.3)) for _ in range(n_frames)]
            with open(os.path.join(occ_dir, 'visible_occupancy.pkl'), 'wb') as f:
                pickle.dump(masks, f)
    meta = dict(categories=categories, DATASET='EmbodiedScan', version='synthetic')
    with open(os.path.join(root, ann_name), 'wb') as f:
        pickle.dump(dict(metainfo=meta, data_list=data_list), f)
    # visual-grounding annotations (mv_3dvg_dataset.py:301-404): single and multiple targets, a target that does not
    # exist in its scan (dropped by the reader), a prompt without target_id (all boxes), rebuilt / given token spans
    import json
    vg = []
    for s, d in enumerate(data_list):
        sid = d['sample_idx']
        noun = class_names[int(scans[s]['gt_labels'][0])]
        t1 = f'find the {noun} that is left of the {class_names[int(scans[s]["gt_labels"][1])]}'
        vg.append(dict(scan_id=sid, text=t1, target_id=1, target=noun, distractor_ids=[2, 3, 4, 5],
                       tokens_positive=[[9, 9 + len(noun)]], anchors=[class_names[int(scans[s]['gt_labels'][1])]], anchor_ids=[2]))
        vg.append(dict(scan_id=sid, text='all the things near the window', target_id=[2, 3], target=['thing', 'thing'],
                       distractor_ids=[], tokens_positive=[[8, 14], [8, 14]]))
        vg.append(dict(scan_id=sid, text='the object that is not there', target_id=n_boxes + 40, target='object',
                       distractor_ids=[], tokens_positive=[[4, 10]]))
        vg.append(dict(scan_id=sid, text='everything in the room'))
        vg.append(dict(scan_id=sid, text='two of them , one missing', target_id=[1, n_boxes + 41], target=['a', 'b'],
                       distractor_ids=[1], tokens_positive=[[0, 3], [15, 18]]))
    with open(os.path.join(root, 'embodiedscan_train_vg.json'), 'w') as f:
        json.dump(vg, f)
    return scans, class_names
