"""The occupancy path (BASELINE config 5) on the CPU, PyTorch f32.  TEST ORACLE.

Functional restatement, driven by a reference-named state dict, of:
  * DenseFusionOccPredictor.extract_feat / loss      models/detectors/dense_fusion_occ.py:120-279
  * AlignedAnchor3DRangeGenerator.anchors_single_range  models/task_modules/anchor/anchor_3d_generator.py:271-354
  * IndoorImVoxelNeck / ResModule                    models/necks/imvoxel_neck.py:34-143
  * ImVoxelOccHead.forward / loss                    models/dense_heads/imvoxel_occ_head.py:73-184
  * occ_multiscale_supervision, geo_scal_loss, sem_scal_loss   models/losses/occ_loss.py:7-141
  * mmdet.FPN (un-vendored; laterals -> nearest top-down -> 3x3 output convs)
Pinned by tests/golden/occ_*.npz, which oracle/make_golden_occ.py records from the reference's OWN occ_loss.py,
imvoxel_neck.py and anchor_3d_generator.py (pure PyTorch, imported from /root/reference)."""
import numpy as np
import torch
import torch.nn.functional as F
from . import coords as C
from . import model as M
from . import rounding as R
from . import sparse as S


# ----------------------------------------------------------------------------- prior grid
def prior_points(n_voxels, anchor_range):
    """grid_anchors([n_voxels[::-1]])[0][:, :3] (dense_fusion_occ.py:156-157): voxel centres, z-major list."""
    fs = n_voxels[::-1]                                   # (D, H, W) = (z, y, x)
    r = torch.tensor(anchor_range)
    z = torch.linspace(r[2], r[5], fs[0] + 1)
    y = torch.linspace(r[1], r[4], fs[1] + 1)
    x = torch.linspace(r[0], r[3], fs[2] + 1)
    z = z + (z[1] - z[0]) / 2
    y = y + (y[1] - y[0]) / 2
    x = x + (x[1] - x[0]) / 2
    gx, gy, gz = torch.meshgrid(x[:fs[2]], y[:fs[1]], z[:fs[0]], indexing='ij')
    ret = torch.stack([gx, gy, gz], -1).permute(2, 1, 0, 3)          # (z, y, x, 3)
    return ret.reshape(-1, 3)


# ----------------------------------------------------------------------------- 2-D neck
def fpn(feats, sd, prefix='neck.'):
    lats = [M._conv2d(f, sd[f'{prefix}lateral_convs.{i}.conv.weight']) + sd[f'{prefix}lateral_convs.{i}.conv.bias'][None, :, None, None]
            for i, f in enumerate(feats)]
    for i in range(len(lats) - 1, 0, -1):
        lats[i - 1] = lats[i - 1] + F.interpolate(lats[i], size=lats[i - 1].shape[2:], mode='nearest')
    return [M._conv2d(l, sd[f'{prefix}fpn_convs.{i}.conv.weight'], 1, 1) + sd[f'{prefix}fpn_convs.{i}.conv.bias'][None, :, None, None]
            for i, l in enumerate(lats)]


# ----------------------------------------------------------------------------- dense 3-D neck
def _conv3d(x, w, stride=1, padding=0):
    return R.op(lambda a, b: F.conv3d(a, b, None, stride, padding), x, w, w.shape[1], w.shape[0])


def _bn3(x, sd, p, training=True):
    return F.batch_norm(x, sd[p + '.running_mean'], sd[p + '.running_var'], sd[p + '.weight'], sd[p + '.bias'], training,
                        0.1, 1e-5)


def _res_module(x, sd, p, stride, training):
    out = F.relu(_bn3(_conv3d(x, sd[p + '.conv1.weight'], stride, 1), sd, p + '.norm1', training))
    out = _bn3(_conv3d(out, sd[p + '.conv2.weight'], 1, 1), sd, p + '.norm2', training)
    idt = x
    if stride != 1:
        idt = _bn3(_conv3d(x, sd[p + '.downsample.0.weight'], stride, 0), sd, p + '.downsample.1', training)
    return F.relu(out + idt)


def imvoxel_neck(x, sd, prefix='neck_3d.', n_blocks=(1, 1, 1), training=True):
    """imvoxel_neck.py:34-58.  x (N, C, X, Y, Z) -> list fine->coarse of (N, out, Xi, Yi, Zi)."""
    n_scales = len(n_blocks)
    down = []
    for i in range(n_scales):
        for b in range(n_blocks[i]):
            x = _res_module(x, sd, f'{prefix}down_layer_{i}.{b}', 2 if (i > 0 and b == 0) else 1, training)
        down.append(x)
    outs = []
    for i in range(n_scales - 1, -1, -1):
        if i < n_scales - 1:
            p = f'{prefix}up_block_{i + 1}'
            x = F.relu(_bn3(R.op(lambda a, b: F.conv_transpose3d(a, b, None, 2), x, sd[p + '.0.weight'], sd[p + '.0.weight'].shape[0], sd[p + '.0.weight'].shape[1]), sd, p + '.1', training))
            x = F.relu(_bn3(_conv3d(x, sd[p + '.3.weight'], 1, 1), sd, p + '.4', training))
            x = down[i] + x
        p = f'{prefix}out_block_{i}'
        outs.append(F.relu(_bn3(_conv3d(x, sd[p + '.0.weight'], 1, 1), sd, p + '.1', training)))
    return outs[::-1]


# ----------------------------------------------------------------------------- losses (occ_loss.py)
def occ_multiscale_supervision(gt_occ, ratio, gt_shape, gt_occupancy_masks=None):
    gt = torch.zeros([gt_shape[0], gt_shape[2], gt_shape[3], gt_shape[4]], dtype=torch.long)
    for i in range(gt.shape[0]):
        coords = torch.div(gt_occ[i][:, :3].long(), ratio, rounding_mode='trunc')
        # duplicate voxels: LAST row wins (our stated convention).  torch's CPU index_put_ only behaves like
        # that while it runs sequentially (small inputs: the golden vectors of the reference's own function); on thousands of
        # rows it is parallelised and the winner is unspecified -- numpy's fancy assignment is sequential by definition
        g = gt[i].numpy()
        c = coords.numpy()
        g[c[:, 0], c[:, 1], c[:, 2]] = gt_occ[i][:, 3].long().numpy()
        if gt_occupancy_masks is not None:
            gt[i][~gt_occupancy_masks[i]] = 255
    return gt


def _bce1(x):
    return F.binary_cross_entropy(x, torch.ones_like(x))


def geo_scal_loss(pred, target):
    p = F.softmax(pred, dim=1)
    empty = p[:, 0]
    nonempty = 1 - empty
    mask = target != 255
    nt = (target != 0)[mask].float()
    nonempty, empty = nonempty[mask], empty[mask]
    eps = 1e-6
    inter = (nt * nonempty).sum()
    precision = inter / (nonempty.sum() + eps)
    recall = inter / (nt.sum() + eps)
    spec = ((1 - nt) * empty).sum() / ((1 - nt).sum() + eps)
    return _bce1(precision) + _bce1(recall) + _bce1(spec)


def sem_scal_loss(pred, target):
    p_all = F.softmax(pred, dim=1)
    loss, count = 0, 0
    mask = target != 255
    for i in range(p_all.shape[1]):
        p = p_all[:, i][mask]
        t = target[mask]
        ct = torch.ones_like(t)
        ct[t != i] = 0
        if torch.sum(ct) > 0:
            count += 1.0
            nom = torch.sum(p * ct)
            lc = 0
            if torch.sum(p) > 0:
                lc = lc + _bce1(nom / torch.sum(p))
            lc = lc + _bce1(nom / torch.sum(ct))
            if torch.sum(1 - ct) > 0:
                lc = lc + _bce1(torch.sum((1 - p) * (1 - ct)) / torch.sum(1 - ct))
            loss = loss + lc
    return loss / count if count else 0 * loss


def head_loss(occ_preds, gt_occupancy, gt_masks=None, return_parts=False):
    """imvoxel_occ_head.py:156-184 (use_semantic=True).  occ_preds: list of (B, C, X, Y, Z) logits."""
    out, parts = {}, []
    for i, pred in enumerate(occ_preds):
        ratio = 2 ** i
        pooled = None
        if gt_masks is not None:
            pooled = [F.max_pool3d(m.float()[None], ratio, stride=ratio)[0].bool() for m in gt_masks]
        gt = occ_multiscale_supervision(gt_occupancy, ratio, pred.shape, pooled)
        ce = F.cross_entropy(pred, gt, ignore_index=255, reduction='mean')
        sem, geo = sem_scal_loss(pred, gt), geo_scal_loss(pred, gt)
        out[f'loss_occ_{i}'] = (ce + sem + geo) * (0.5 ** i)
        parts.append((ce, sem, geo, gt))
    return (out, parts) if return_parts else out


# ----------------------------------------------------------------------------- detector
def voxelize_range(points_list, range_min, voxel_size, clamp_max):
    """dense_fusion_occ.py:227-245: (p - min) / voxel_size in f32, truncated into int32 (ME sparse_collate), clamped;
    unique voxels keep their first point; root set in Z-curve order (same free choice as oracle.coords.voxelize)."""
    cs = []
    rmin = np.asarray(range_min, np.float32)
    vs = np.asarray(voxel_size, np.float32)
    for b, p in enumerate(points_list):
        p = np.asarray(p, np.float32)
        q = ((p[:, :3] - rmin[None]).astype(np.float32) / vs[None]).astype(np.float32)
        c = np.trunc(q).astype(np.int32)
        c = np.minimum(np.maximum(c, 0), np.asarray(clamp_max, np.int32)[None])
        cs.append(np.concatenate([np.full((c.shape[0], 1), b, np.int32), c], 1))
    c = np.concatenate(cs, 0)
    uk, first, _ = C.unique_first(C.pack(c))
    order = np.argsort(C.morton_key(c[first]), kind='stable')
    first = first[order]
    return c[first], first


def detector_forward(sd, points, imgs, metas, n_voxels, point_cloud_range, prior_range, n_blocks=(1, 1, 1), training=True):
    """DenseFusionOccPredictor.extract_feat + ImVoxelOccHead.forward -> list of (1, C, Xi, Yi, Zi) logits."""
    B, V = imgs.shape[:2]
    assert B == 1
    feats = M.resnet50_w16(imgs.reshape((-1,) + imgs.shape[2:]), sd)        # widths come from the state dict
    x = fpn(feats, sd)[0]
    x = x.reshape((B, V) + x.shape[1:])
    prior = prior_points(list(n_voxels), prior_range)
    meta = metas[0]
    if 'origin' in meta['depth2img']:
        prior = prior + torch.as_tensor(meta['depth2img']['origin'], dtype=torch.float32)
    sf = torch.tensor(meta['scale_factor'][:2], dtype=torch.float32) if 'scale_factor' in meta else 1
    off = torch.tensor(meta['img_crop_offset'], dtype=torch.float32) if 'img_crop_offset' in meta else 0
    vol = M.batch_point_sample(meta, x[0], prior, M.projection_matrices(meta), sf, off, meta.get('flip', False),
                               imgs.shape[-2:], meta['img_shape'][:2])
    img_volume = vol.reshape(list(n_voxels[::-1]) + [-1]).permute(3, 2, 1, 0)[None]        # (1, C, X, Y, Z)
    stride = 64
    vs = [(prior_range[3 + a] - prior_range[a]) / n_voxels[a] / stride for a in range(3)]
    cmax = [n * stride - 1 for n in n_voxels]
    coords, src = voxelize_range([p.detach().numpy() for p in points], point_cloud_range[:3], vs, cmax)
    f = torch.cat([p[:, :3] for p in points])[torch.from_numpy(src)]
    last = M.mink_resnet34(S.SpT(coords, f, 1, 1, {}), sd, training=training)[-1]
    X, Y, Z = n_voxels
    dense = f.new_zeros((X * Y * Z, last.feats.shape[1]))
    c = torch.from_numpy(last.coords[:, 1:].astype(np.int64)) // stride
    dense = dense.index_copy(0, (c[:, 0] * Y + c[:, 1]) * Z + c[:, 2], last.feats)
    point_volume = dense.reshape(X, Y, Z, -1).permute(3, 0, 1, 2)[None]
    x3 = imvoxel_neck(torch.cat([img_volume, point_volume], 1), sd, n_blocks=n_blocks, training=training)
    return [_conv3d(l, sd[f'bbox_head.occ.{i}.weight'], 1, 0) for i, l in enumerate(x3)]


def detector_loss(sd, points, imgs, metas, gt_occupancy, gt_masks, n_voxels, point_cloud_range, prior_range,
                  n_blocks=(1, 1, 1), return_aux=False):
    preds = detector_forward(sd, points, imgs, metas, n_voxels, point_cloud_range, prior_range, n_blocks)
    losses, parts = head_loss(preds, gt_occupancy, gt_masks, return_parts=True)
    return (losses, dict(preds=preds, parts=parts)) if return_aux else losses
