"""Device driver of the multi-view point<->image fusion (row A8): packs the per-sample meta the
reference reads from `img_meta` (point_fusion.py:48-70, sparse_featfusion_single_stage.py:144-164) into one
small device block and launches the fused es_point_sample kernels once per level for the WHOLE batch
(the reference loops sample x level and replicates the points V times)."""
import numpy as np
import torch
from .... import hip
from ....hip import CONSTS, P, call

_OP = {'T': 1, 'S': 2, 'R': 3, 'HF': 4, 'VF': 5}


def _stream():
    return hip.stream()


def build_fusion_meta(metas, coord_type, img_pad_shape, n_views):
    """-> (B, 32 + 16*V) float32 host tensor following the ES_FUSE_* layout of include/es_hip.h."""
    stride = CONSTS['ES_FUSE_PROJ'] + 16 * n_views
    out = torch.zeros((len(metas), stride), dtype=torch.float32)
    key = {'LIDAR': 'lidar2img', 'DEPTH': 'depth2img', 'CAMERA': 'cam2img'}[coord_type.upper()]
    for b, m in enumerate(metas):
        row = out[b]
        flow = list(m.get('transformation_3d_flow', []))[::-1]
        hf, vf = m.get('pcd_horizontal_flip', False), m.get('pcd_vertical_flip', False)
        ops = [o for o in flow if not ((o == 'HF' and not hf) or (o == 'VF' and not vf))]
        assert len(ops) <= 8
        row[CONSTS['ES_FUSE_NOPS']] = len(ops)
        for i, o in enumerate(ops):
            row[CONSTS['ES_FUSE_OPS'] + i] = _OP[o]
        rot = torch.tensor(np.asarray(m['pcd_rotation']), dtype=torch.float32) if 'pcd_rotation' in m else torch.eye(3)
        row[CONSTS['ES_FUSE_ROTINV']:CONSTS['ES_FUSE_ROTINV'] + 9] = rot.inverse().reshape(-1)
        row[CONSTS['ES_FUSE_ISCALE']] = 1.0 / m.get('pcd_scale_factor', 1.)
        tr = torch.tensor(np.asarray(m['pcd_trans']), dtype=torch.float32) if 'pcd_trans' in m else torch.zeros(3)
        row[CONSTS['ES_FUSE_NTRANS']:CONSTS['ES_FUSE_NTRANS'] + 3] = -tr
        sf = m.get('scale_factor', (1., 1.))
        row[CONSTS['ES_FUSE_SFX']], row[CONSTS['ES_FUSE_SFY']] = float(sf[0]), float(sf[1])
        off = m.get('img_crop_offset', (0., 0.))
        row[CONSTS['ES_FUSE_CROPX']], row[CONSTS['ES_FUSE_CROPY']] = float(off[0]), float(off[1])
        row[CONSTS['ES_FUSE_FLIP']] = 1.0 if m.get('flip', False) else 0.0
        row[CONSTS['ES_FUSE_ORIW']] = float(m['img_shape'][1])
        row[CONSTS['ES_FUSE_PADW']], row[CONSTS['ES_FUSE_PADH']] = float(img_pad_shape[1]), float(img_pad_shape[0])
        pm = m[key]
        assert isinstance(pm, dict) and isinstance(pm['intrinsic'], list) and len(pm['extrinsic']) == n_views
        for v in range(n_views):
            intr = torch.tensor(np.asarray(pm['intrinsic'][v]), dtype=torch.float32)
            extr = torch.tensor(np.asarray(pm['extrinsic'][v]), dtype=torch.float32)
            row[CONSTS['ES_FUSE_PROJ'] + 16 * v:CONSTS['ES_FUSE_PROJ'] + 16 * (v + 1)] = (intr @ extr).reshape(-1)
    return out


def batch_point_sample_level(cs, voxel_size, meta_dev, n_views, feat, Hf, Wf, out, col0):
    """Writes the sampled image features of every voxel of `cs` into out[:, col0:col0+C] and registers the
    backward (deterministic gather into the feature-map gradient).  feat: Var ((B*V*Hf*Wf), C)."""
    C = feat.d.shape[1]
    n = cs.n
    pix = torch.empty((n, n_views), dtype=torch.int32, device=out.device)
    cnt = torch.empty(n, dtype=torch.int32, device=out.device)
    # (feature maps stored in bf16 by the image backbone's activation storage: the _h variant widens them while summing)
    call('es_point_sample_fwd_h' if feat.d.dtype == torch.bfloat16 else 'es_point_sample_fwd', P(cs.coords), n, float(voxel_size),
         P(meta_dev), meta_dev.shape[1], n_views, P(feat.d), Hf, Wf, C, out.data_ptr() + 4 * col0, out.stride(0), P(pix), P(cnt),
         _stream())
    return pix, cnt


def batch_point_sample_level_bwd(cs, n_views, dout, col0, pix, cnt, feat, Hf, Wf):
    C = feat.d.shape[1]
    if not feat.rg:
        return
    acc = 1
    if feat.g is None:
        feat.g, acc = torch.empty(feat.d.shape, dtype=torch.float32, device=feat.d.device), 0   # the gather writes every pixel
    n_pix = feat.d.shape[0]
    head = torch.empty(n_pix, dtype=torch.int32, device=feat.d.device)
    nxt = torch.empty(max(cs.n * n_views, 1), dtype=torch.int32, device=feat.d.device)
    call('es_point_sample_bwd', P(cs.coords), cs.n, n_views, dout.data_ptr() + 4 * col0, dout.stride(0), P(pix), P(cnt),
         Hf, Wf, C, P(feat.g), n_pix // (Hf * Wf), P(head), P(nxt), acc, _stream())
