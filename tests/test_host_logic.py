"""CPU-side checks (no GPU): the C-ABI library loads and exports every symbol the header declares, the host-side
packing / parameter / config logic, and the data-parallel exchange on 2 gloo ranks."""
import os
import pytest
import subprocess
import sys
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from embodiedscan_amd import hip
    import ctypes
    lib = ctypes.CDLL(hip.LIB_PATH)
    assert len(hip.PROTOS) >= 40
    for name, (ret, argtypes, argnames) in hip.PROTOS.items():
        assert hasattr(lib, name), name
        assert len(argtypes) == len(argnames)
    # the header is the single source of truth for the per-sample fusion meta block
    assert hip.CONSTS['ES_FUSE_PROJ'] == 32 and hip.CONSTS['ES_MAX_SEG'] == 32      # batch 12 (mv-grounding 8xb12) needs > 8 instance-norm segments


def test_param_arena_reference_names_and_roundtrip():
    from embodiedscan_amd.params import ParamArena, detector_specs
    a = ParamArena(detector_specs(), seed=3)
    sd = a.state_dict()
    # reference state-dict names / shapes (mmdet.ResNet, MinkResNet, FCAF3DHeadRotMat)
    assert sd['backbone.conv1.weight'].shape == (16, 3, 7, 7)
    assert sd['backbone.layer4.2.conv3.weight'].shape == (512, 128, 1, 1)
    assert sd['backbone_3d.conv1.kernel'].shape == (27, 3, 64)
    assert sd['backbone_3d.layer4.0.downsample.0.kernel'].shape == (256, 512)
    assert sd['bbox_head.up_block_3.0.kernel'].shape == (8, 1024, 512)
    assert sd['bbox_head.conv_cls.kernel'].shape == (128, 284) and sd['bbox_head.conv_cls.bias'].shape == (1, 284)
    assert sd['bbox_head.conv_center.kernel'].shape == (128, 1) and sd['bbox_head.conv_reg.kernel'].shape == (128, 12)
    assert abs(float(sd['bbox_head.conv_cls.bias'][0, 0]) + 4.59512) < 1e-4
    n3d = sum(v.numel() for k, v in sd.items() if k.startswith('backbone_3d.') and 'running' not in k)
    assert n3d == 63457088                      # MinkResNet34 parameter count (63.46 M)
    b = ParamArena(detector_specs(), seed=9)
    b.load_state_dict(sd)
    assert all(torch.equal(a.p[k], b.p[k]) for k in a.p)
    train = set(a.trainable_names())
    assert 'backbone.layer1.0.conv1.weight' not in train and 'backbone.layer2.0.conv1.weight' in train
    assert not any(k.endswith('bn1.weight') for k in train if k.startswith('backbone.'))   # frozen 2-D BN


def test_config_loader_and_registry():
    from embodiedscan_amd.config import load_config
    from embodiedscan_amd.registry import MODELS
    import embodiedscan_amd.models  # noqa: F401
    cfg = load_config(os.path.join(ROOT, 'configs', 'mv_3ddet.py'))
    assert cfg['model']['type'] == 'SparseFeatureFusionSingleStage3DDetector'
    for t in ('mmdet.ResNet', 'MinkResNet', 'FCAF3DHeadRotMat', 'Det3DDataPreprocessor', cfg['model']['type']):
        assert MODELS.get(t) is not None
    head = MODELS.build(dict(cfg['model']['bbox_head'], train_cfg=None, test_cfg=None))
    assert head.decouple_weights == [0.2, 0.2, 0.2, 0.4] and head.pts_center_threshold == 18


def test_fusion_meta_matches_reference_order_of_ops():
    """the reverse 3-D augmentation op list and projection matrices are packed as the reference applies them
    (point_fusion.py:79-105, sparse_featfusion_single_stage.py:160-164)."""
    from embodiedscan_amd.hip import CONSTS
    from embodiedscan_amd.models.layers.fusion_layers.point_fusion import build_fusion_meta
    from embodiedscan_amd.synth import make_scan
    from oracle import model as OM
    s = make_scan(5, n_views=2, height=60, width=80, img_size=(64, 64), n_points=500, n_boxes=3)
    m = build_fusion_meta([s['meta']], 'DEPTH', (64, 64), 2)[0]
    flow = s['meta']['transformation_3d_flow'][::-1]
    codes = [{'T': 1, 'S': 2, 'R': 3, 'HF': 4, 'VF': 5}[o] for o in flow]
    assert int(m[CONSTS['ES_FUSE_NOPS']]) == len(codes)
    assert [int(v) for v in m[CONSTS['ES_FUSE_OPS']:CONSTS['ES_FUSE_OPS'] + len(codes)]] == codes
    proj = OM.projection_matrices(s['meta'])
    np.testing.assert_array_equal(m[CONSTS['ES_FUSE_PROJ']:CONSTS['ES_FUSE_PROJ'] + 32].reshape(2, 4, 4).numpy(), proj.numpy())
    assert float(m[CONSTS['ES_FUSE_SFX']]) == float(np.float32(64 / 80)) and float(m[CONSTS['ES_FUSE_PADW']]) == 64


def test_synthetic_scan_is_seeded_and_shaped():
    from embodiedscan_amd.synth import make_scan
    a = make_scan(42, n_views=2, height=60, width=80, img_size=(64, 64), n_points=1000, n_boxes=4)
    b = make_scan(42, n_views=2, height=60, width=80, img_size=(64, 64), n_points=1000, n_boxes=4)
    c = make_scan(43, n_views=2, height=60, width=80, img_size=(64, 64), n_points=1000, n_boxes=4)
    assert np.array_equal(a['depth'], b['depth']) and np.array_equal(a['sel_pix'], b['sel_pix'])
    assert not np.array_equal(a['depth'], c['depth'])
    assert a['depth'].shape == (2, 60, 80) and a['img'].shape == (2, 3, 64, 64) and a['gt_boxes'].shape == (4, 9)
    assert (a['depth'].reshape(2, -1)[a['sel_view'], a['sel_pix']] > 0).all()      # PointSample draws non-zero depth


def test_oracle_coordinate_rules():
    """edge cases of the voxel rules the kernels are held to: truncation toward zero, first point wins,
    empty input, strided floor for negatives, generative children, union order."""
    from oracle import coords as C
    p = np.array([[0.005, -0.005, 0.0], [-0.0149, 0.0149, 0.02], [0.0051, -0.0001, 0.0099], [0.031, 0.0, -0.031]], np.float32)
    c, src = C.voxelize([p], 0.01)
    assert sorted(map(tuple, c.tolist())) == sorted([(0, 0, 0, 0), (0, -1, 1, 2), (0, 3, 0, -3)])
    assert src[[tuple(r) for r in c.tolist()].index((0, 0, 0, 0))] == 0          # first occurrence survives
    e, es = C.voxelize([np.zeros((0, 3), np.float32)], 0.01)
    assert e.shape == (0, 4) and es.shape == (0,)
    s = C.stride_coords(np.array([[0, -1, 1, 2], [0, -2, 0, 3], [0, 3, 0, -3]], np.int32), 2)
    assert s.tolist() == [[0, -2, 0, 2], [0, 2, 0, -4]]
    g = C.gen_transpose_coords(np.array([[0, 4, 0, -4]], np.int32), 4)
    assert g.shape == (8, 4) and g[7].tolist() == [0, 6, 2, -2]
    a = np.array([[0, 0, 0, 0], [1, 0, 0, 0]], np.int32)
    b = np.array([[0, 2, 0, 0], [0, 0, 0, 0], [1, 2, 0, 0]], np.int32)
    u, pa, pb = C.union_coords(a, b, 2)
    assert u.tolist() == [[0, 0, 0, 0], [0, 2, 0, 0], [1, 0, 0, 0], [1, 2, 0, 0]] and pb.tolist() == [1, 0, 3]


def _dp_worker():
    import torch.distributed as dist
    from embodiedscan_amd.parallel import allreduce_mean_, reduce_mean
    from embodiedscan_amd.params import ParamArena, fcaf3d_head_specs
    dist.init_process_group('gloo')
    rank, world = dist.get_rank(), dist.get_world_size()
    arena = ParamArena(fcaf3d_head_specs(in_channels=(8, 16), out_channels=8, n_classes=5), seed=0)   # same seed: replicas
    chk = arena.data.clone()
    dist.broadcast(chk, 0)
    assert torch.equal(chk, arena.data), 'replicas must start identical'
    g = torch.Generator().manual_seed(100 + rank)
    arena.grad.copy_(torch.randn(arena.n_train, generator=g))
    mine = arena.grad.clone()
    allreduce_mean_(arena.grad)
    others = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(others, mine)
    assert torch.allclose(arena.grad, torch.stack(others).mean(0), atol=1e-7)
    n_pos = torch.tensor([10.0 + rank, 0.0, 3.0 * rank])            # per-sample positive counts of this rank
    avg = reduce_mean(n_pos).clamp(min=1.0)
    assert torch.allclose(avg, torch.tensor([10.5, 1.0, 1.5]))
    assert torch.equal(n_pos, torch.tensor([10.0 + rank, 0.0, 3.0 * rank]))   # not modified in place
    # bucketed, overlapped variant: buckets are launched in backward-completion order (head, 3-D backbone, 2-D backbone)
    from embodiedscan_amd.parallel import BucketedGradReducer
    from embodiedscan_amd.params import detector_specs
    big = ParamArena([s for s in detector_specs(n_classes=5) if ('layer' not in s.name or '.0.' in s.name)], seed=0)
    red = BucketedGradReducer(big)
    assert red.ranges[0][0] == 0 and red.ranges[-1][1] == big.n_train
    assert all(red.ranges[i][1] == red.ranges[i + 1][0] for i in range(2))
    g2 = torch.Generator().manual_seed(7 + rank)
    big.grad.copy_(torch.randn(big.n_train, generator=g2))
    mine = big.grad.clone()
    for part in (2, 1, 0):
        red.launch(part)
    assert red.finish() == 3
    red.scale_()                                                 # (on a GPU the 1/world is folded into the AdamW pass)
    gathered = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(gathered, mine)
    assert torch.allclose(big.grad, torch.stack(gathered).mean(0), atol=1e-7)
    assert not red.finish()                                      # nothing pending any more
    # the grounder's four parts (backbones, MinkNeck, decoder + head + text map) and the occupancy detector's (image branch,
    # MinkResNet, fine neck, coarse neck + head) with the 256 MB chunking forced down to 1 MB: every element is reduced
    # exactly once whatever the launch order of the parts, and no single call carries more than the chunk bound
    import embodiedscan_amd.parallel as PAR
    from embodiedscan_amd.models.detectors.dense_fusion_occ import DenseFusionOccPredictor
    from embodiedscan_amd.models.detectors.sparse_featfusion_grounder import SparseFeatureFusion3DGrounder
    from embodiedscan_amd.params import grounder_specs, occ_detector_specs
    PAR.MAX_BUCKET_FLOATS = 1 << 18
    cases = (('grounder', grounder_specs(text_dim=32, E=64, num_layers=1, ffn=64), SparseFeatureFusion3DGrounder._bucket_groups, (3, 2, 1, 0)),
             ('occupancy', occ_detector_specs(base_channels=8, fpn_out=16, neck_in=16 + 512, neck_out=16, n_blocks=[1, 1, 1], num_classes=5,
                                              head_in=[16, 16, 16]), DenseFusionOccPredictor._bucket_groups, (3, 2, 1, 0)))
    for name, specs, groups, order in cases:
        ar = ParamArena(specs, seed=0)
        red = BucketedGradReducer(ar, groups=groups)
        assert len(red.parts) == len(groups) + 1 and all(red.parts), (name, [len(p) for p in red.parts])
        sizes = [sum(b - a for a, b in red.chunks(k)) for k in range(len(red.parts))]
        assert sum(sizes) == ar.n_train and max(b - a for k in range(len(red.parts)) for a, b in red.chunks(k)) <= PAR.MAX_BUCKET_FLOATS
        gk = torch.Generator().manual_seed(31 + rank)
        ar.grad.copy_(torch.randn(ar.n_train, generator=gk))
        mine = ar.grad.clone()
        n_calls = 0
        for part in order:
            n_calls += len(red.chunks(part))
            red.launch(part)
        assert red.finish() == n_calls
        red.scale_()
        gathered = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine)
        assert torch.allclose(ar.grad, torch.stack(gathered).mean(0), atol=1e-6), name
        if rank == 0:
            print(f'{name}: parts {[s * 4 // 2 ** 20 for s in sizes]} MiB in {n_calls} all-reduce calls')
    # N4 / 8e: every rank reads ITS shard of the scan list from files -- same number of batches on both ranks, the shards
    # are the two halves of one seeded permutation (no collective in the data path; this gather is the test's own)
    from embodiedscan_amd.datasets import EmbodiedScanDataset, ScanLoader
    from embodiedscan_amd.synth import _NOUNS
    fix = os.path.join(ROOT, 'tests', 'golden', 'fake_dataset')
    pipe = [dict(type='LoadAnnotations3D'),
            dict(type='MultiViewPipeline', n_images=3,
                 transforms=[dict(type='LoadImageFromFile'), dict(type='LoadDepthFromFile'),
                             dict(type='ConvertRGBDToPoints', coord_type='CAMERA'), dict(type='PointSample', num_points=200),
                             dict(type='Resize', scale=(48, 48), keep_ratio=False)]),
            dict(type='AggregateMultiViewPoints', coord_type='DEPTH'), dict(type='PointSample', num_points=500),
            dict(type='Pack3DDetInputs', keys=['img', 'points', 'gt_bboxes_3d', 'gt_labels_3d'])]
    ds = EmbodiedScanDataset(fix, 'embodiedscan_infos_train.pkl', metainfo=dict(classes=_NOUNS + ['object']), pipeline=pipe)
    loader = ScanLoader(ds, batch_size=2, rank=rank, world=world, shuffle=True, seed=5, times=6, num_threads=2, pin=False)
    mine_ids = [s['meta']['scan_id'] for b in loader for s in b]
    both = [None] * world
    dist.all_gather_object(both, (len(loader), loader.indices(), mine_ids))
    assert both[0][0] == both[1][0] == 3 and len(mine_ids) == 6
    import torch as _t
    g3 = _t.Generator()
    g3.manual_seed(5)
    perm = [i % len(ds) for i in _t.randperm(len(ds) * 6, generator=g3).tolist()]
    assert both[0][1] == perm[0::2] and both[1][1] == perm[1::2]
    assert mine_ids == [ds.get_data_info(i)['scan_id'] for i in loader.indices()[:6]]
    dist.destroy_process_group()
    print(f'rank {rank} ok')


def test_data_parallel_exchange_two_gloo_ranks():
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', PYTHONPATH=ROOT)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2', '--master-addr', '127.0.0.1',
           '--master-port', '29611', os.path.abspath(__file__), '--dp-worker']
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.count('ok') == 2


if __name__ == '__main__' and '--dp-worker' in sys.argv:
    sys.path.insert(0, ROOT)
    _dp_worker()


def test_checkpoint_roundtrip_reference_layout(tmp_path):
    """N3: a checkpoint is an mmengine-style file whose state_dict uses the reference's names / shapes; weights and the
    AdamW moments survive save -> load into a fresh arena bit for bit; missing / unexpected keys are reported."""
    import torch
    from embodiedscan_amd.checkpoint import load_checkpoint, save_checkpoint
    from embodiedscan_amd.optim import OptimWrapper
    from embodiedscan_amd.params import ParamArena, detector_specs
    a = ParamArena(detector_specs(284), seed=3)
    opt = OptimWrapper(lr=2e-3)
    opt.m = torch.randn(a.n_train)
    opt.v = torch.rand(a.n_train)
    opt.step = 17
    path = save_checkpoint(a, str(tmp_path / 'ck.pth'), optim=opt, meta=dict(epoch=5))
    raw = torch.load(path, weights_only=False)
    sd = raw['state_dict']
    assert sd['backbone.layer2.0.conv2.weight'].shape == (32, 32, 3, 3)           # torch (O, I, KH, KW)
    assert sd['backbone_3d.layer1.0.conv1.kernel'].shape == (27, 64, 64)          # ME (K, I, O)
    assert sd['bbox_head.conv_cls.kernel'].shape == (128, 284) and sd['bbox_head.conv_reg.kernel'].shape == (128, 12)
    # nn.BatchNorm buffers a strict load on the reference side expects; resume state under its own (non-mmengine) key
    assert int(sd['backbone_3d.layer1.0.norm1.bn.num_batches_tracked']) == 17 and int(sd['backbone.bn1.num_batches_tracked']) == 0
    assert 'optimizer' not in raw and 'optimizer_flat' in raw
    b = ParamArena(detector_specs(284), seed=99)
    opt2 = OptimWrapper()
    opt2.m, opt2.v = torch.zeros(b.n_train), torch.zeros(b.n_train)
    missing, unexpected, meta = load_checkpoint(b, path, optim=opt2)
    assert not missing and not unexpected and meta['epoch'] == 5
    # compared in the reference's view: arena padding and the 13 never-trained centre / regression bias slots of the
    # fused head GEMM are not part of the reference's state
    sa, sb = a.state_dict(), b.state_dict()
    assert list(sa) == list(sb) and all(torch.equal(sa[k], sb[k]) for k in sa)
    for x, y in ((opt.m, opt2.m), (opt.v, opt2.v)):
        ra, rb = a.flat_to_ref(x), b.flat_to_ref(y)
        assert list(ra) == list(rb) and all(torch.equal(ra[k], rb[k]) for k in ra)
    assert opt2.step == 17 and opt2.lr == 2e-3
    # partial / foreign checkpoints
    del sd['bbox_head.conv_cls.bias']
    sd['module.extra.weight'] = torch.zeros(3)
    torch.save(dict(state_dict={'module.' + k if not k.startswith('module.') else k: v for k, v in sd.items()}),
               str(tmp_path / 'ddp.pth'))
    missing, unexpected, _ = load_checkpoint(ParamArena(detector_specs(284), seed=1), str(tmp_path / 'ddp.pth'))
    assert missing == ['bbox_head.conv_cls.bias'] and unexpected == ['extra.weight']


def test_gt_box_augmentation_matches_reference(golden_dir):
    """product host function pipeline.augment_gt_boxes against the golden vectors made by the reference's box class"""
    import torch
    from embodiedscan_amd.pipeline import augment_gt_boxes
    for name in ('augment_hv', 'augment_h', 'augment_none'):
        d = np.load(os.path.join(golden_dir, name + '.npz'))
        aug = dict(hflip=bool(d['hflip']), vflip=bool(d['vflip']), rot=d['rot_mat_T'], scale=float(d['scale']), trans=d['trans'])
        b = augment_gt_boxes(torch.from_numpy(d['boxes']), aug).numpy()
        np.testing.assert_allclose(b[:, :6], d['boxes_out'][:, :6], rtol=2e-6, atol=2e-6)
        da = (b[:, 6:] - d['boxes_out'][:, 6:] + np.pi) % (2 * np.pi) - np.pi
        assert np.abs(da).max() < 5e-6
    assert augment_gt_boxes(torch.zeros((0, 9)), aug).shape == (0, 9)


def test_lr_schedule_matches_torch_multistep():
    """param_scheduler of the reference config (MultiStepLR, milestones [8, 11], gamma 0.1, 12 epochs) against
    torch.optim.lr_scheduler.MultiStepLR stepped once per epoch"""
    import torch
    from embodiedscan_amd.config import build_optim_wrapper, build_param_scheduler, load_config
    cfg = load_config(os.path.join(ROOT, 'configs', 'mv_3ddet.py'))
    ow = build_optim_wrapper(cfg)
    sched, = build_param_scheduler(cfg, ow)
    p = torch.nn.Parameter(torch.zeros(1))
    topt = torch.optim.AdamW([p], lr=ow.lr)
    tsch = torch.optim.lr_scheduler.MultiStepLR(topt, milestones=[8, 11], gamma=0.1)
    for epoch in range(12):
        assert abs(ow.lr - topt.param_groups[0]['lr']) < 1e-12, (epoch, ow.lr)
        topt.step(); tsch.step(); sched.step()
    assert abs(ow.lr - 1e-5) < 1e-12


def test_resume_continues_lr_schedule(tmp_path):
    """save at epoch 9 (lr already decayed to 1e-4), resume into fresh objects: the lr must stay 1e-4, the
    next epochs must follow the original schedule (no reset to 1e-3, no double decay), whatever the construction order."""
    from embodiedscan_amd.checkpoint import load_checkpoint, save_checkpoint
    from embodiedscan_amd.config import build_optim_wrapper, build_param_scheduler, load_config
    from embodiedscan_amd.params import ParamArena, detector_specs
    cfg = load_config(os.path.join(ROOT, 'configs', 'mv_3ddet.py'))
    a = ParamArena(detector_specs(284), seed=3)
    ow = build_optim_wrapper(cfg)
    sch = build_param_scheduler(cfg, ow)
    expect = []
    for epoch in range(12):
        if epoch == 9:
            save_checkpoint(a, str(tmp_path / 'e9.pth'), optim=ow, schedulers=sch, meta=dict(epoch=9))
        expect.append(ow.lr)
        sch[0].step()
    assert abs(expect[9] - 1e-4) < 1e-12
    for order in ('sched_before_load', 'sched_after_load'):
        b = ParamArena(detector_specs(284), seed=4)
        ow2 = build_optim_wrapper(cfg)
        if order == 'sched_before_load':
            sch2 = build_param_scheduler(cfg, ow2)
            load_checkpoint(b, str(tmp_path / 'e9.pth'), optim=ow2, schedulers=sch2)
        else:
            load_checkpoint(b, str(tmp_path / 'e9.pth'), optim=ow2)
            sch2 = build_param_scheduler(cfg, ow2)              # base_lr comes from the config, not the decayed live lr
            load_checkpoint(b, str(tmp_path / 'e9.pth'), schedulers=sch2)
        got = []
        for epoch in range(9, 12):
            got.append(ow2.lr)
            sch2[0].step()
        assert all(abs(g - e) < 1e-12 for g, e in zip(got, expect[9:])), (order, got, expect[9:])


def test_upload_gts_layout():
    """one packed buffer for the ground truth of a batch: per-sample views carry boxes, R(-euler) and int32 labels"""
    import torch
    from embodiedscan_amd.models.dense_heads.fcaf3d_head import upload_gts
    from oracle import geometry as G
    g = torch.Generator().manual_seed(5)
    gts = [(torch.randn(n, 9, generator=g), torch.randint(0, 284, (n,), generator=g)) for n in (3, 0, 5)]
    out = upload_gts(gts, torch.device('cpu'))
    assert [o[0].shape[0] for o in out] == [3, 0, 5]
    for (b, l), (db, dr, dl) in zip(gts, out):
        assert torch.equal(db, b) and dl.dtype == torch.int32 and torch.equal(dl.long(), l)
        if len(b):
            assert torch.allclose(dr.view(-1, 3, 3), G.euler_to_matrix_zxy(-b[:, 6:9]), atol=1e-7)
    assert upload_gts([(torch.zeros(0, 9), torch.zeros(0, dtype=torch.long))], torch.device('cpu'))[0][0].shape == (0, 9)


def test_gradient_buckets_tile_every_full_detector():
    """for the FULL parameter sets of all three detectors the overlap buckets are contiguous and tile
    [0, n_train) exactly (nothing is reduced twice, nothing is left out -- the grounder's neck / decoder / text map live
    in the last bucket)."""
    from embodiedscan_amd.parallel import BucketedGradReducer
    from embodiedscan_amd.params import ParamArena, detector_specs, grounder_specs, occ_detector_specs

    class Lazy(ParamArena):                 # offsets only: no 750 M-float allocation in a CPU test
        def __init__(self, specs):
            self.specs = specs
            order = [s for s in specs if s.trainable] + [s for s in specs if not s.trainable]
            off, self.offsets = 0, {}
            for s in order:
                n = int(np.prod(s.shape)) if s.shape else 1
                self.offsets[s.name] = (off, n)
                off += (n + 3) // 4 * 4
                if s.trainable:
                    self.n_train = off
            self.total = off
    from embodiedscan_amd.models.detectors.dense_fusion_occ import DenseFusionOccPredictor
    from embodiedscan_amd.models.detectors.sparse_featfusion_grounder import SparseFeatureFusion3DGrounder
    from embodiedscan_amd.models.detectors.sparse_featfusion_single_stage import SparseFeatureFusionSingleStage3DDetector
    from embodiedscan_amd.parallel import MAX_BUCKET_FLOATS
    for name, specs, cls in (('mv-3ddet', detector_specs(284), SparseFeatureFusionSingleStage3DDetector),
                             ('grounder', grounder_specs(), SparseFeatureFusion3DGrounder),
                             ('occupancy', occ_detector_specs(), DenseFusionOccPredictor)):
        a = Lazy(specs)
        red = BucketedGradReducer(a, groups=cls._bucket_groups)
        assert len(red.parts) == len(cls._bucket_groups) + 1 and all(red.parts), name
        spans = sorted(r for k in range(len(red.parts)) for r in red.chunks(k))
        assert spans[0][0] == 0 and spans[-1][1] == a.n_train and all(x[1] == y[0] for x, y in zip(spans[:-1], spans[1:])), name
        covered = sum(b - a_ for a_, b in spans)
        assert covered == a.n_train and max(b - a_ for a_, b in spans) <= MAX_BUCKET_FLOATS, (name, covered, a.n_train)
        part_mib = [sum(b - a_ for a_, b in red.chunks(k)) * 4 // 2 ** 20 for k in range(len(red.parts))]
        print(f'{name}: parts {part_mib} MiB in {len(spans)} all-reduce calls of <= {MAX_BUCKET_FLOATS * 4 // 2 ** 20} MiB tile [0, {a.n_train})')
        if name == 'occupancy':                 # the 2.9 GB of neck gradients are NOT one call at the end of backward
            assert part_mib[-1] > 2000 and len(red.chunks(len(red.parts) - 1)) >= 8 and part_mib[0] < 200


def test_reduce_mean_called_once_per_step(monkeypatch):
    """the per-sample positive counts of a whole batch travel in ONE collective (the reference issues one per sample):
    count the calls the FCAF3D loss makes through parallel.reduce_mean on a CPU stand-in of its phase 2"""
    import inspect
    from embodiedscan_amd.models.dense_heads import fcaf3d_head, grounding_head
    for mod, fn in ((fcaf3d_head.FCAF3DHeadRotMat.loss_by_levels, 'reduce_mean('), (grounding_head.GroundingHead.loss, 'reduce_mean(')):
        src = '\n'.join(l.split('#')[0] for l in inspect.getsource(mod).splitlines())      # comments stripped
        assert src.count(fn) == 1, f'{mod.__qualname__} must reduce the positive counts exactly once per step'
        body = src.split(fn)[0]
        assert 'for ' not in body.split('\n')[-1], 'the collective must not sit inside a per-sample loop'


def test_paramwise_lr_groups_follow_the_grounding_config():
    """configs/grounding/...py:196-201: decoder lr x0.1, text encoder frozen -> contiguous AdamW ranges"""
    from embodiedscan_amd.optim import OptimWrapper
    from embodiedscan_amd.params import ParamArena, grounder_specs
    a = ParamArena.__new__(ParamArena)
    specs = grounder_specs()
    a.specs = specs
    order = [s for s in specs if s.trainable] + [s for s in specs if not s.trainable]
    off, a.offsets = 0, {}
    for s in order:
        n = int(np.prod(s.shape)) if s.shape else 1
        a.offsets[s.name] = (off, n)
        off += (n + 3) // 4 * 4
        if s.trainable:
            a.n_train = off
    ow = OptimWrapper(lr=5e-4, paramwise={'decoder': dict(lr_mult=0.1, decay_mult=1.0)})
    ow._build_groups(a)
    g = ow.groups
    assert g[0][0] == 0 and g[-1][1] == a.n_train and all(x[1] == y[0] for x, y in zip(g[:-1], g[1:]))
    assert [x[2] for x in g] == [1.0, 0.1, 1.0]                      # [backbones + neck | decoder | text map + head]
    dec = [a.offsets[n] for n in a.trainable_names() if n.startswith('decoder.')]
    assert g[1][0] == min(o for o, _ in dec) and g[1][1] == max(o + (n + 3) // 4 * 4 for o, n in dec)


def test_hash_tokenizer_and_positive_map():
    """the tokenizer stand-in honours the protocol create_positive_map relies on (sparse_featfusion_grounder.py:570-621)"""
    from embodiedscan_amd.text import HashTokenizer, create_positive_map
    tk = HashTokenizer()
    texts = ['find the red chair near the table', 'the lamp']
    enc = tk.batch_encode_plus(texts, padding='longest', return_tensors='pt')
    assert enc.input_ids.shape == (2, 9) and enc.attention_mask.sum(1).tolist() == [9, 4]
    assert enc.input_ids[0, 0] == 0 and enc.input_ids[1, 3] == 2 and enc.input_ids[1, 4] == 1        # <s> ... </s> <pad>
    beg = texts[0].index('red')
    pm = create_positive_map(enc, [[[beg, beg + len('red chair')]]], 0, max_num_entities=16)
    assert pm.shape == (1, 16) and pm[0].nonzero().squeeze(1).tolist() == [3, 4]                      # tokens of "red chair"
    assert abs(float(pm.sum()) - 1.0) < 1e-5
    same = tk.batch_encode_plus(texts).input_ids
    assert (same == enc.input_ids).all()


def test_checkpoint_tap_order_permutation():
    """N3 hardening: a checkpoint whose sparse kernels number the 27 (and 8) offsets differently is re-ordered while
    loading; a synthetic z-fastest / arbitrarily permuted state dict reproduces the original arena bit for bit, and the
    sparse convolution it drives is unchanged (checked on the CPU oracle)."""
    import torch
    from embodiedscan_amd.params import ParamArena, fcaf3d_head_specs, mink_resnet34_specs
    from oracle import coords as C, sparse as S
    specs = mink_resnet34_specs() + fcaf3d_head_specs(in_channels=(8, 16), out_channels=8, n_classes=5)
    a = ParamArena(specs, seed=11)
    sd = a.state_dict()
    g = torch.Generator().manual_seed(0)
    for order in ('z_fastest', {27: torch.randperm(27, generator=g).tolist(), 8: torch.randperm(8, generator=g).tolist()}):
        p27, p8 = (ParamArena.tap_permutation(order.get(27) if isinstance(order, dict) else order, 3),
                   ParamArena.tap_permutation(order.get(8) if isinstance(order, dict) else order, 2))
        foreign = {}
        for k, v in sd.items():                      # write the checkpoint in the foreign numbering: ck[p[k]] = ours[k]
            if k.endswith('.kernel') and v.dim() == 3 and v.shape[0] in (27, 8):
                p = p27 if v.shape[0] == 27 else p8
                w = torch.empty_like(v)
                w[torch.tensor(p)] = v
                foreign[k] = w
            else:
                foreign[k] = v
        b = ParamArena(specs, seed=99)
        missing, unexpected = b.load_state_dict(foreign, tap_order=order)
        assert not missing and not unexpected and torch.equal(a.data, b.data)
        c = ParamArena(specs, seed=99)
        c.load_state_dict(foreign)                   # without the hook the taps are silently permuted
        assert not torch.equal(a.data, c.data)
    assert ParamArena.tap_permutation('z_fastest', 3)[1] == 9 and ParamArena.tap_permutation(None, 2) == list(range(8))
    # the semantic claim behind the hook: permuting taps AND offsets together leaves the convolution unchanged
    pts = np.random.default_rng(0).uniform(-1, 1, (300, 3)).astype(np.float32)
    co, _ = C.voxelize([pts], 0.1)
    x = torch.randn(co.shape[0], 4, generator=g)
    w = torch.randn(27, 4, 6, generator=g)
    y = S.conv(S.SpT(co, x, 1, 1, {}), w, 3).feats
    nbr = C.kernel_map(co, co, 3, 1)
    perm = ParamArena.tap_permutation('z_fastest', 3)
    y2 = S.gather_conv(x, nbr[:, perm], w[torch.tensor(perm)])
    assert torch.allclose(y, y2, atol=1e-6)


def test_grounder_state_dict_carries_the_text_encoder(tmp_path):
    """`text_encoder.*` (the frozen RoBERTa, a submodule of the reference detector) is written by state_dict() /
    save_checkpoint and absorbed by load_state_dict; a strict load of a dict without those keys reports them missing
    (round-2 advisor finding: they were silently dropped and the prompts encoded by random weights)."""
    import torch
    from embodiedscan_amd.checkpoint import load_checkpoint, save_checkpoint
    from embodiedscan_amd.config import build_detector, load_config
    cfg = load_config(os.path.join(ROOT, 'configs', 'mv_grounding.py'))
    cfg['model']['text_encoder_cfg'] = dict(hidden_size=32, num_hidden_layers=1, num_attention_heads=2, intermediate_size=64,
                                            max_position_embeddings=40, vocab_size=50265)
    a = build_detector(cfg, device='cpu', seed=0)
    sd = a.state_dict()
    tkeys = [k for k in sd if k.startswith('text_encoder.')]
    assert 'text_encoder.embeddings.word_embeddings.weight' in tkeys and any('encoder.layer.0' in k for k in tkeys)
    f = save_checkpoint(a, str(tmp_path / 'g.pth'))
    b = build_detector(cfg, device='cpu', seed=5)                       # different random text encoder and arena
    w0 = b.text_encoder.embeddings.word_embeddings.weight.clone()
    assert not torch.equal(w0, a.text_encoder.embeddings.word_embeddings.weight)
    missing, unexpected, _ = load_checkpoint(b, f, strict=True)
    assert not missing and not unexpected
    for (ka, va), (kb, vb) in zip(a.text_encoder.state_dict().items(), b.text_encoder.state_dict().items()):
        assert ka == kb and torch.equal(va, vb), ka
    assert torch.equal(a.arena.data[:a.arena.n_train], b.arena.data[:b.arena.n_train])
    bare = {k: v for k, v in sd.items() if not k.startswith('text_encoder.')}
    missing, unexpected = b.load_state_dict(bare, strict=False)
    assert any(k.startswith('text_encoder.') for k in missing) and not unexpected
    with pytest.raises(RuntimeError):
        b.load_state_dict(bare, strict=True)
    missing, unexpected = b.load_state_dict(dict(sd, **{'text_encoder.bogus.weight': torch.zeros(1)}), strict=False)
    assert unexpected == ['text_encoder.bogus.weight']


def test_effective_cpus_honours_the_cgroup_quota(tmp_path):
    """the GPU boxes show 256 logical CPUs and grant 16 cores (cpu.max = '1600000 100000'): pools are sized by the grant"""
    import os
    from embodiedscan_amd.datasets.loader import effective_cpus
    have = len(os.sched_getaffinity(0))
    assert effective_cpus(str(tmp_path / 'absent')) == have
    v2 = tmp_path / 'v2'
    v2.mkdir()
    (v2 / 'cpu.max').write_text('max 100000\n')
    assert effective_cpus(str(v2)) == have
    (v2 / 'cpu.max').write_text('150000 100000\n')
    assert effective_cpus(str(v2)) == 1
    (v2 / 'cpu.max').write_text('1600000 100000\n')
    assert effective_cpus(str(v2)) == min(have, 16)
    v1 = tmp_path / 'v1'
    (v1 / 'cpu').mkdir(parents=True)
    (v1 / 'cpu' / 'cpu.cfs_quota_us').write_text('-1\n')
    (v1 / 'cpu' / 'cpu.cfs_period_us').write_text('100000\n')
    assert effective_cpus(str(v1)) == have
    (v1 / 'cpu' / 'cpu.cfs_quota_us').write_text('300000\n')
    assert effective_cpus(str(v1)) == min(have, 3)
    (v2 / 'cpu.max').write_text('garbage\n')
    assert effective_cpus(str(v2)) == have


def test_bench_self_launches_its_ranks_when_no_launcher_is_around():
    """`python bench.py --gpus 2` with WORLD_SIZE unset re-executes itself under torch.distributed.run (two local ranks, rendezvous on
    127.0.0.1, a free port) instead of dying on a WORLD_SIZE assertion: the dry-launch mode runs that path end to end without a GPU
    (process group on gloo, one all-reduce, ONE JSON line from rank 0); a mismatching launcher is refused with the command to use."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env['ES_DIST_BACKEND'] = 'gloo'
    bench = os.path.join(ROOT, 'bench.py')
    r = subprocess.run([sys.executable, bench, '--gpus', '2', '--dry-launch'], capture_output=True, text=True, timeout=300, env=env, cwd='/tmp')
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['ranks'] == 2 and out['allreduce_ok'] and out['master'].startswith('127.0.0.1:')
    r = subprocess.run([sys.executable, bench, '--gpus', '2', '--dry-launch'], capture_output=True, text=True, timeout=120,
                       env=dict(env, WORLD_SIZE='1', RANK='0'), cwd='/tmp')
    assert r.returncode != 0 and 'torch.distributed.run' in r.stderr


def test_host_thread_pool_is_capped_by_the_granted_cores(monkeypatch):
    """engine.settle_host_threads (first train step of a process): torch's intra-op pool goes down to a quarter of the cores the
    cgroup grants per local rank, never up; ES_HOST_THREADS overrides; 0 leaves torch alone (profiles/r5w_*: 128 OpenMP threads on a
    16-core quota got the step loop throttled)"""
    import torch
    from embodiedscan_amd import engine as E
    from embodiedscan_amd.datasets import loader
    have = torch.get_num_threads()
    calls = []
    monkeypatch.setattr(torch, 'set_num_threads', lambda n: calls.append(n))
    try:
        for granted, local, env, seen, want in ((16, None, None, 128, [4]), (16, '8', None, 128, [1]), (16, None, '2', 128, [2]),
                                                (16, None, '0', 128, []), (64, None, None, 8, []), (2, None, None, 128, [1])):
            calls.clear()
            monkeypatch.setattr(loader, 'effective_cpus', lambda g=granted: g)
            monkeypatch.setattr(torch, 'get_num_threads', lambda s=seen: s)
            for k, v in (('LOCAL_WORLD_SIZE', local), ('ES_HOST_THREADS', env)):
                if v is None:
                    monkeypatch.delenv(k, raising=False)
                else:
                    monkeypatch.setenv(k, v)
            E.settle_host_threads(force=True)
            assert calls == want, (granted, local, env, seen, calls)
        calls.clear()
        E.settle_host_threads()                       # once per process
        assert calls == []
    finally:
        monkeypatch.undo()
        assert torch.get_num_threads() == have
        E._HOST_SETTLED[0] = False
