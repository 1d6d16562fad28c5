# Model section of the mv-3ddet configuration (values follow the reference's
# configs/detection/mv-det3d_8xb4_embodiedscan-3d-284class-9dof.py:17-58, which can also be passed
# to embodiedscan_amd.config.load_config unchanged) and its data section (:134-200, read by
# embodiedscan_amd.config.build_dataloader -> datasets.EmbodiedScanDataset / ScanLoader).  Runtime sections are out of scope.
n_points = 100000
n_views = 20
model = dict(
    type='SparseFeatureFusionSingleStage3DDetector',
    data_preprocessor=dict(type='Det3DDataPreprocessor', mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375],
                           bgr_to_rgb=True, pad_size_divisor=32),
    backbone=dict(type='mmdet.ResNet', depth=50, base_channels=16, num_stages=4, out_indices=(0, 1, 2, 3),
                  frozen_stages=1, norm_cfg=dict(type='BN', requires_grad=False), norm_eval=True, style='pytorch'),
    backbone_3d=dict(type='MinkResNet', in_channels=3, depth=34),
    use_xyz_feat=True,
    bbox_head=dict(type='FCAF3DHeadRotMat', in_channels=(128, 256, 512, 1024), out_channels=128, voxel_size=.01,
                   pts_prune_threshold=100000, pts_assign_threshold=27, pts_center_threshold=18, num_classes=284,
                   num_reg_outs=12, center_loss=dict(type='mmdet.CrossEntropyLoss', use_sigmoid=True),
                   bbox_loss=dict(type='BBoxCDLoss', mode='l1', loss_weight=1.0, group='g8'),
                   cls_loss=dict(type='mmdet.FocalLoss'), decouple_bbox_loss=True, decouple_groups=4,
                   decouple_weights=[0.2, 0.2, 0.2, 0.4]),
    coord_type='DEPTH', train_cfg=dict(), test_cfg=dict(nms_pre=1000, iou_thr=.5, score_thr=.01))
optim_wrapper = dict(type='OptimWrapper', optimizer=dict(type='AdamW', lr=0.001, weight_decay=0.0001),
                     clip_grad=dict(max_norm=10, norm_type=2))
dataset_type = 'EmbodiedScanDataset'
data_root = 'data'
train_pipeline = [
    dict(type='LoadAnnotations3D'),
    dict(type='MultiViewPipeline', n_images=n_views,
         transforms=[dict(type='LoadImageFromFile'), dict(type='LoadDepthFromFile'),
                     dict(type='ConvertRGBDToPoints', coord_type='CAMERA'),
                     dict(type='PointSample', num_points=n_points // 10),
                     dict(type='Resize', scale=(480, 480), keep_ratio=False)]),
    dict(type='AggregateMultiViewPoints', coord_type='DEPTH'),
    dict(type='PointSample', num_points=n_points),
    dict(type='RandomFlip3D', sync_2d=False, flip_2d=False, flip_ratio_bev_horizontal=0.5, flip_ratio_bev_vertical=0.5),
    dict(type='GlobalRotScaleTrans', rot_range=[-0.087266, 0.087266], scale_ratio_range=[.9, 1.1],
         translation_std=[.1, .1, .1], shift_height=False),
    dict(type='Pack3DDetInputs', keys=['img', 'points', 'gt_bboxes_3d', 'gt_labels_3d'])]
test_pipeline = [
    dict(type='LoadAnnotations3D'),
    dict(type='MultiViewPipeline', n_images=50, ordered=True,
         transforms=[dict(type='LoadImageFromFile'), dict(type='LoadDepthFromFile'),
                     dict(type='ConvertRGBDToPoints', coord_type='CAMERA'),
                     dict(type='PointSample', num_points=n_points // 10),
                     dict(type='Resize', scale=(480, 480), keep_ratio=False)]),
    dict(type='AggregateMultiViewPoints', coord_type='DEPTH'),
    dict(type='PointSample', num_points=n_points),
    dict(type='Pack3DDetInputs', keys=['img', 'points', 'gt_bboxes_3d', 'gt_labels_3d'])]
train_dataloader = dict(batch_size=4, num_workers=4, sampler=dict(type='DefaultSampler', shuffle=True),
                        dataset=dict(type='RepeatDataset', times=10,
                                     dataset=dict(type=dataset_type, data_root=data_root,
                                                  ann_file='embodiedscan_infos_train.pkl', pipeline=train_pipeline,
                                                  test_mode=False, filter_empty_gt=True, box_type_3d='Euler-Depth')))
val_dataloader = dict(batch_size=1, num_workers=1, sampler=dict(type='DefaultSampler', shuffle=False),
                      dataset=dict(type=dataset_type, data_root=data_root, ann_file='embodiedscan_infos_val.pkl',
                                   pipeline=test_pipeline, test_mode=True, filter_empty_gt=True, box_type_3d='Euler-Depth'))
# reference config :214,225-230: 12 epochs, learning rate x0.1 after epochs 8 and 11
train_cfg = dict(type='EpochBasedTrainLoop', max_epochs=12, val_interval=1)
param_scheduler = dict(type='MultiStepLR', begin=0, end=12, by_epoch=True, milestones=[8, 11], gamma=0.1)
