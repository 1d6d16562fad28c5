"""Generate tests/golden/occ_*.npz by running the REFERENCE's own pure-PyTorch occupancy code.

Run in the build container only (needs /root/reference):   python -m oracle.make_golden_occ
TEST INFRASTRUCTURE (see oracle/make_golden.py for the mechanism).  Reference entry points exercised (file:line):
  occ_multiscale_supervision / geo_scal_loss / sem_scal_loss     models/losses/occ_loss.py:7-141
  IndoorImVoxelNeck.forward (+ ResModule)                        models/necks/imvoxel_neck.py:34-143
  AlignedAnchor3DRangeGenerator.grid_anchors                     models/task_modules/anchor/anchor_3d_generator.py:94-137,271-354
"""
import os
import numpy as np
import torch


def main(out_dir=None):
    from . import _ref_stubs
    _ref_stubs.install()
    from embodiedscan.models.losses.occ_loss import geo_scal_loss, occ_multiscale_supervision, sem_scal_loss
    from embodiedscan.models.necks.imvoxel_neck import IndoorImVoxelNeck
    from embodiedscan.models.task_modules.anchor.anchor_3d_generator import AlignedAnchor3DRangeGenerator
    out_dir = out_dir or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')
    g = torch.Generator().manual_seed(20240925)

    # ---- prior grid of the shipped config (configs/occupancy/mv-occ...py:9-11,52) and a small one
    rng = [-3.2, -3.2, -1.28, 3.2, 3.2, 1.28]
    gen = AlignedAnchor3DRangeGenerator(ranges=[rng], rotations=[.0])
    a_full = gen.grid_anchors([[16, 40, 40]], device='cpu')[0]
    a_small = gen.grid_anchors([[4, 6, 8]], device='cpu')[0]
    np.savez_compressed(os.path.join(out_dir, 'occ_anchors.npz'), range=np.array(rng, np.float32),
                        full_xyz=a_full[:, :3].numpy(), full_rest=a_full[:3, 3:].numpy(), small=a_small.numpy())

    # ---- supervision + the three losses (values and gradients) on a small volume with ignored voxels
    C, X, Y, Z = 11, 8, 6, 4
    pred = torch.randn(1, C, X, Y, Z, generator=g) * 2
    n = 90
    occ = torch.stack([torch.randint(0, X, (n,), generator=g), torch.randint(0, Y, (n,), generator=g),
                       torch.randint(0, Z, (n,), generator=g), torch.randint(1, C, (n,), generator=g)], 1)
    occ[5:12, 3] = 0                                        # explicit empties overwrite earlier labels
    mask = torch.rand(X, Y, Z, generator=g) > 0.2
    rec = dict(pred=pred.numpy(), gt_occ=occ.numpy(), mask=mask.numpy())
    for ratio, shape in ((1, (1, C, X, Y, Z)), (2, (1, C, X // 2, Y // 2, Z // 2))):
        pooled = torch.nn.MaxPool3d(ratio, stride=ratio)(mask.float()[None])[0].bool()
        rec[f'gt_r{ratio}_masked'] = occ_multiscale_supervision([occ], ratio, shape, [pooled]).numpy()
        rec[f'gt_r{ratio}'] = occ_multiscale_supervision([occ], ratio, shape, None).numpy()
    for tag, gt in (('masked', torch.from_numpy(rec['gt_r1_masked'])), ('plain', torch.from_numpy(rec['gt_r1']))):
        p = pred.clone().requires_grad_(True)
        ce = torch.nn.CrossEntropyLoss(ignore_index=255, reduction='mean')(p, gt.long())
        sem = sem_scal_loss(p, gt.long())
        geo = geo_scal_loss(p, gt.long())
        (ce + sem + geo).backward()
        rec[f'ce_{tag}'], rec[f'sem_{tag}'], rec[f'geo_{tag}'] = ce.item(), sem.item(), geo.item()
        rec[f'grad_{tag}'] = p.grad.numpy()
    np.savez_compressed(os.path.join(out_dir, 'occ_loss.npz'), **rec)

    # ---- IndoorImVoxelNeck, small widths, train-mode BN, forward + input gradient
    torch.manual_seed(7)
    neck = IndoorImVoxelNeck(in_channels=8, out_channels=4, n_blocks=[1, 1, 1])
    for m in neck.modules():
        if isinstance(m, torch.nn.BatchNorm3d):
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.uniform_(-0.2, 0.2)
    neck.train()
    x = torch.randn(1, 8, 8, 8, 4, generator=g).requires_grad_(True)
    sd = {k: v.detach().clone().numpy() for k, v in neck.state_dict().items() if 'num_batches' not in k}
    outs = neck(x)
    sum((o * o).sum() for o in outs).backward()
    rec = {'sd.' + k: v for k, v in sd.items()}
    rec.update(x=x.detach().numpy(), dx=x.grad.numpy(), **{f'out{i}': o.detach().numpy() for i, o in enumerate(outs)})
    rec['dw_conv1'] = neck.down_layer_0[0].conv1.weight.grad.numpy()
    rec['dw_up'] = neck.up_block_1[0].weight.grad.numpy()
    np.savez_compressed(os.path.join(out_dir, 'occ_neck.npz'), **rec)
    print('wrote occ_anchors.npz occ_loss.npz occ_neck.npz to', out_dir)


if __name__ == '__main__':
    main()
