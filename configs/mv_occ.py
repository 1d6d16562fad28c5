# Model section of the multi-view occupancy configuration (values follow the reference's
# configs/occupancy/mv-occ_8xb1_embodiedscan-occ-80class.py:1-53, which embodiedscan_amd.config.load_config also reads
# unchanged).  Dataset / runtime sections are out of scope.
n_points = 100000
point_cloud_range = [-3.2, -3.2, -0.78, 3.2, 3.2, 1.78]
prior_generator = dict(type='AlignedAnchor3DRangeGenerator', ranges=[[-3.2, -3.2, -1.28, 3.2, 3.2, 1.28]], rotations=[.0])
model = dict(
    type='DenseFusionOccPredictor', use_valid_mask=False, use_xyz_feat=True, point_cloud_range=point_cloud_range,
    data_preprocessor=dict(type='Det3DDataPreprocessor', mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375],
                           bgr_to_rgb=True, pad_size_divisor=32),
    backbone=dict(type='mmdet.ResNet', depth=50, num_stages=4, out_indices=(0, 1, 2, 3), frozen_stages=1,
                  norm_cfg=dict(type='BN', requires_grad=False), norm_eval=True, style='pytorch'),
    neck=dict(type='mmdet.FPN', in_channels=[256, 512, 1024, 2048], out_channels=256, num_outs=4),
    backbone_3d=dict(type='MinkResNet', in_channels=3, depth=34),
    neck_3d=dict(type='IndoorImVoxelNeck', in_channels=256 + 512, out_channels=128, n_blocks=[1, 1, 1]),
    bbox_head=dict(type='ImVoxelOccHead', volume_h=[20, 10, 5], volume_w=[20, 10, 5], volume_z=[8, 4, 2], num_classes=81,
                   in_channels=[128, 128, 128], use_semantic=True),
    prior_generator=prior_generator, n_voxels=[40, 40, 16], coord_type='DEPTH')
optim_wrapper = dict(type='OptimWrapper', optimizer=dict(type='AdamW', lr=0.0001, weight_decay=0.01),
                     clip_grad=dict(max_norm=35., norm_type=2))
# data section of the reference config (configs/occupancy/mv-occ_8xb1_embodiedscan-occ-80class.py:76-132); `metainfo`
# (classes = occ_classes = the 80 names, :42-72) is passed by the caller
n_points = 100000
train_pipeline = [
    dict(type='LoadAnnotations3D', with_occupancy=True, with_visible_occupancy_masks=True),
    dict(type='MultiViewPipeline', n_images=10,
         transforms=[dict(type='LoadImageFromFile'), dict(type='LoadDepthFromFile'),
                     dict(type='ConvertRGBDToPoints', coord_type='CAMERA'),
                     dict(type='PointSample', num_points=n_points // 10),
                     dict(type='Resize', scale=(480, 480), keep_ratio=False)]),
    dict(type='AggregateMultiViewPoints', coord_type='DEPTH'),
    dict(type='PointsRangeFilter', point_cloud_range=[-3.2, -3.2, -0.78, 3.2, 3.2, 1.78]),
    dict(type='PointSample', num_points=n_points),
    dict(type='ConstructMultiViewMasks'),
    dict(type='Pack3DDetInputs', keys=['img', 'points', 'gt_bboxes_3d', 'gt_labels_3d', 'gt_occupancy'])]
train_dataloader = dict(batch_size=1, num_workers=1, sampler=dict(type='DefaultSampler', shuffle=True),
                        dataset=dict(type='EmbodiedScanDataset', data_root='data', ann_file='embodiedscan_infos_train.pkl',
                                     pipeline=train_pipeline, test_mode=False, filter_empty_gt=True,
                                     box_type_3d='Euler-Depth'))
