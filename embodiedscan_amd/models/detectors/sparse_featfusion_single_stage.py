"""SparseFeatureFusionSingleStage3DDetector on the MI355X kernels.

Same registry name, constructor arguments, `forward(inputs, data_samples, mode)` protocol and loss keys as
embodiedscan/models/detectors/sparse_featfusion_single_stage.py:28-330; `train_step` follows mmengine's
BaseModel.train_step (preprocess -> forward(mode='loss') -> parse_losses -> optimiser update).
"""
import os
import torch
from ... import hip
from ... import engine as E
from ... import sparse
from ...hip import P, call
from ...parallel import BucketedGradReducer, is_dist
from ...params import ParamArena, detector_specs
from ...registry import MODELS
from ...sparse import SparseTensor
from ..layers.fusion_layers.point_fusion import (batch_point_sample_level, batch_point_sample_level_bwd,
                                                 build_fusion_meta)


def _stream():
    return hip.stream()


@MODELS.register_module()
class SparseFeatureFusionSingleStage3DDetector:
    _version = 2

    def __init__(self, backbone, backbone_3d, bbox_head, neck=None, neck_3d=None, coord_type='CAMERA',
                 train_cfg=None, test_cfg=None, data_preprocessor=None, use_xyz_feat=False, init_cfg=None, seed=0,
                 device='cuda:0'):
        assert neck is None and neck_3d is None, 'the shipped mv-3ddet config has no necks'
        self.device = torch.device(device)
        self.backbone = MODELS.build(backbone)
        self.backbone.act16 = True              # its feature maps only feed the projection fusion: bf16 activation storage
        self.backbone_3d = MODELS.build(backbone_3d)
        bbox_head = dict(bbox_head)
        bbox_head.update(train_cfg=train_cfg, test_cfg=test_cfg)
        self.bbox_head = MODELS.build(bbox_head)
        self.data_preprocessor = MODELS.build(data_preprocessor, device=self.device) if data_preprocessor else None
        self.voxel_size = self.bbox_head.voxel_size
        self.use_xyz_feat = use_xyz_feat
        self.coord_type = coord_type
        self.train_cfg, self.test_cfg = train_cfg, test_cfg
        self.arena = ParamArena(detector_specs(self.bbox_head.num_classes), seed=seed)
        self.training = True
        self._bound = False
        self._pf_init()
        # The image branch (2-D backbone forward, and its backward) and the point branch (coordinate pipeline + 3-D
        # backbone) only meet in the fusion layer, so they are issued on two HIP streams (engine.side_stream): the deep,
        # under-filled sparse launches of the point branch then run in the shadow of the image branch's full-chip ones.

    # ------------------------------------------------------------------ parameters
    def to(self, device):
        self.device = torch.device(device)
        self.arena.to(self.device)
        if self.data_preprocessor is not None:
            self.data_preprocessor.to(self.device)
        self._bound = False
        return self

    def _bind(self):
        if not self._bound:
            if self.arena.data.device != self.device:
                self.arena.to(self.device)
            E.begin_bind(id(self))               # conv kernels of an earlier bind of this detector are dropped
            try:
                self.backbone.bind(self.arena, 'backbone.')
                self.backbone_3d.bind(self.arena, 'backbone_3d.')
                self.bbox_head.bind(self.arena, 'bbox_head.')
            finally:
                E.end_bind()
            self._bound = True

    def __del__(self):
        try:
            E.release(id(self))
        except Exception:
            pass

    def state_dict(self):
        return self.arena.state_dict()

    def load_state_dict(self, sd, strict=False, tap_order=None):
        """reference-named state dict -> arena; returns (missing, unexpected) keys like torch.nn.Module does"""
        res = self.arena.load_state_dict(sd, strict=strict, tap_order=tap_order)
        E.WEIGHT_VERSION[0] += 1                 # bf16 weight copies are stale
        if self._bound:
            self.backbone.refresh()              # re-fold the frozen BatchNorm2d statistics
        return res

    def train(self, mode=True):
        self.training = mode
        self.backbone_3d.training = mode
        self.bbox_head.training = mode
        return self

    # ------------------------------------------------------------------ next-batch prefetch (round 4)
    # The reference hides the data side of step i+1 behind step i with DataLoader workers (A1-A3 on host cores).  Here A1-A4 and
    # the coordinate manager's work (A6: strided sets, kernel maps, unions) run on the GPU, and everything in them is
    # WEIGHT-INDEPENDENT: depth -> points, image normalisation, voxel keys, Morton order, the strided chain with its row-count
    # read-backs, the backbone's kernel / inverse maps, the head's unions.  prefetch() issues that prefix for the NEXT batch on a
    # high-priority side stream right after train_step() of the current batch has been queued: its kernels (~2 ms, small
    # grids) and its host round trips run under the current step's backward pass, and the next train_step() starts with the
    # 3-D backbone's first convolution instead of 4.5 ms of coordinate work with the chip a quarter busy
    # (profiles/r3v_timeline.txt).  Nothing that depends on the weights is touched (the image backbone of step i+1 must see
    # the weights AFTER step i's update), so the step computes exactly what it computed before (tests/test_gpu_prefetch.py:
    # bit-identical losses and gradients with and without).
    # Allocator discipline: tensors made here come from the prefetch stream's pool but are read by main-stream kernels of the
    # next step.  They are kept referenced until the prefetch stream has waited for an event recorded on the main stream
    # at the end of the step that used them -- one step late, when that event has long fired -- so a recycled block can never
    # be rewritten by a prefetch kernel while a main-stream kernel still reads it.
    def _pf_init(self):
        self._pf_stream = None
        self._prefetched = None        # dict of the batch prepared ahead (consumed by the next train_step on the same `data`)
        self._pf_cur = None            # ... of the batch the running train_step is using
        self._pf_hold = []             # [(event at the end of the step, prefetched dict)] awaiting release

    def prefetch(self, make_data):
        """data = make_data() (e.g. pipeline.make_batch of the next scans: A1-A3 are launched on the prefetch stream) + the
        weight-independent prefix of train_step(data, ...) issued now.  Returns `data`; hand exactly this object to the next
        train_step().  Optional: train_step() on any other data works as before."""
        if self._pf_stream is None:
            self._pf_stream = torch.cuda.Stream(device=self.device, priority=int(os.environ.get('ES_PF_PRIORITY', '-1')))
        st = self._pf_stream
        while len(self._pf_hold) > 1:                          # the step before the last one has long finished on the device
            ev, old = self._pf_hold.pop(0)
            st.wait_event(ev)
            del old
        # The prefetch is a chain of ~200 small launches with a handful of host round trips -- and so is the 3-D backbone's forward
        # pass.  Two latency-bound chains interleaved on the chip slow each other several-fold (measured, profiles/r4g_*: 3-D
        # forward 3.2 -> 10 ms), so the prefetch kernels are held back until the step queued last has finished that stage and
        # the main stream runs the head's large launches.
        gate = getattr(self, '_ev_3d_done', None)
        if gate is not None and os.environ.get('ES_PF_GATE', '1') != '0':
            st.wait_event(gate)
        else:
            # no gate (first step, or ES_PF_GATE=0): the preprocessor's two image buffers are only safe to overwrite once the step
            # queued last has read its own -- wait for everything it has queued, image branch included
            st.wait_stream(hip.stream_obj())
            if E._SIDE:
                st.wait_event(E._SIDE['join'])
        self._bind()
        try:
            with torch.cuda.stream(st):
                hip.refresh_stream()
                data = make_data()
                pre = self.data_preprocessor(data, True) if self.data_preprocessor is not None else data
                pts = self._points_f32(pre['inputs']['points'])
                cs, src = sparse.voxelize(pts, self.voxel_size)
                self._prefetch_coords(cs, maps=True)
                ready = torch.cuda.Event()
                ready.record(st)
        finally:
            hip.refresh_stream()                               # launches follow torch's current stream again
        self._prefetched = dict(data=data, pre=pre, pts=pts, cs=cs, src=src, ready=ready)
        return data

    @staticmethod
    def _points_f32(points):
        return [p if (p.dtype == torch.float32 and p.stride(-1) == 1) else p.float().contiguous() for p in points]

    def _prefetch_coords(self, cs, maps=False):
        """all data-dependent row counts of the point branch (strided sets, unions of the head's top-down pass) are read back
        here; maps=True (next-batch prefetch): also build the 3-D backbone's kernel / inverse maps now"""
        lv_sets = self.backbone_3d.prefetch_coords(cs)
        if maps and hasattr(self.backbone_3d, 'prefetch_maps'):
            self.backbone_3d.prefetch_maps(cs)
        for m in (getattr(self, 'bbox_head', None), getattr(self, 'neck_3d', None)):
            if m is not None and hasattr(m, 'prefetch_coords'):
                m.prefetch_coords(lv_sets, maps=maps)
                break

    def _pf_take(self, data):
        """train_step entry: the prefetched batch if `data` is the object prefetch() returned, else None"""
        pf, self._prefetched = self._prefetched, None
        if pf is not None and pf['data'] is data:
            hip.stream_obj().wait_event(pf['ready'])
            self._pf_cur = pf
            return pf
        self._pf_cur = None
        return None

    def _pf_done(self):
        """train_step exit: the batch's prefetched tensors stay referenced until the prefetch stream has seen this event"""
        if self._pf_cur is not None:
            ev = torch.cuda.Event()
            ev.record(hip.stream_obj())
            self._pf_hold.append((ev, self._pf_cur))
            self._pf_cur = None

    # ------------------------------------------------------------------ features
    def extract_feat(self, batch_inputs_dict, batch_data_samples):
        """sparse_featfusion_single_stage.py:86-221.  Returns 4 SparseTensors with [3-D | image] channels."""
        self._bind()
        # image features first: views folded into the batch dimension (:130-136), channels-last row matrices.  The 2-D
        # backbone needs no host round trip, so its ~10 ms of kernels are queued BEFORE the coordinate pipeline, whose
        # data-dependent row counts force a few stream synchronisations -- those then overlap with the queued work.
        img = batch_inputs_dict['imgs']
        B, V = img.shape[:2]
        H, W = img.shape[-2:]
        if img.stride(2) != 1:       # (B,V,3,H,W) given NCHW-contiguous: convert once to channels-last
            img = img.permute(0, 1, 3, 4, 2).contiguous().permute(0, 1, 4, 2, 3)
        nhwc = img.permute(0, 1, 3, 4, 2).reshape(B * V, H, W, 3)
        forked = E.TWO_STREAMS[0]
        E.mark('A18 preprocess + weight cast' if not forked else 'A18 preprocess')
        if forked:
            E.refresh_weight_copies()            # one cast launch for all kernels, before the branches split
            with E.side_stream():
                img_feats = self.backbone(nhwc)
        else:
            E.refresh_weight_copies()
            img_feats = self.backbone(nhwc)
        E.mark('A7 2-D backbone fwd')
        self._tape_marks = [len(E.TAPE.fns)]                   # end of the 2-D backbone's closures
        # the projection meta table (pure host arithmetic on the samples' matrices, 1-2 ms for 4 x 20 views) is built HERE, while the
        # device works through the image backbone and before the host waits on the coordinate phase's row counts -- round 4 built it
        # behind the 3-D backbone, where the main stream had run dry: a 0.84 ms hole in every step (profiles/r5i_critical_chain.txt)
        metas = [ds.metainfo for ds in batch_data_samples]
        meta_dev = build_fusion_meta(metas, self.coord_type, (H, W), V).pin_memory().to(self.device, non_blocking=True)
        pf = self._pf_cur
        if pf is not None and batch_inputs_dict is pf['pre']['inputs']:
            pts, cs, src = pf['pts'], pf['cs'], pf['src']      # voxelised (and mapped) under the previous step's backward
        else:
            pts = self._points_f32(batch_inputs_dict['points'])
            cs, src = sparse.voxelize(pts, self.voxel_size)
        # voxel features (:109-116): the whole point row (xyz + extra columns) with use_xyz_feat, else the columns behind xyz
        c0 = 0 if self.use_xyz_feat else 3
        Cf = int(pts[0].shape[1]) - c0
        assert Cf > 0, 'use_xyz_feat=False needs point columns beyond xyz'
        allp = torch.cat([p[:, c0:] for p in pts]) if len(pts) > 1 else pts[0][:, c0:].contiguous()
        feats = torch.empty((cs.n, Cf), dtype=torch.float32, device=allp.device)
        call('es_row_move', P(feats), Cf, P(allp), allp.stride(0), P(src), cs.n, Cf, 0, _stream())
        # all data-dependent row counts of the point branch (strided sets, unions of the head's top-down pass) are read back
        # here, under the image branch's kernels; the 3-D backbone and the head are then queued without host stalls
        if os.environ.get('ES_PREFETCH_COORDS', '1') != '0':
            self._prefetch_coords(cs)                          # (everything is cached already after a next-batch prefetch)
        E.mark('A4 voxelise')
        x = self.backbone_3d(SparseTensor(cs, E.Var(feats, rg=False)))
        E.mark('A5+A6 3-D backbone fwd + maps')
        if self._pf_stream is not None:                        # next-batch prefetch in use: its kernels start behind this point
            self._ev_3d_done = torch.cuda.Event()
            self._ev_3d_done.record(hip.stream_obj())
        self._tape_marks.append(len(E.TAPE.fns))               # end of the 3-D backbone's closures
        if forked:
            E.join_side()                        # image features are needed from here on
        outs = []
        for lvl, xl in enumerate(x):
            f2d, Hf, Wf = img_feats[lvl]
            C3, C2 = xl.F.d.shape[1], f2d.d.shape[1]
            cat = torch.empty((xl.cs.n, C3 + C2), dtype=torch.float32, device=self.device)
            E.copy_cols(cat, 0, xl.F.d)
            pix, cnt = batch_point_sample_level(xl.cs, self.voxel_size, meta_dev, V, f2d, Hf, Wf, cat, C3)
            y = E.Var(cat)

            def bwd(y=y, xl=xl, f2d=f2d, Hf=Hf, Wf=Wf, pix=pix, cnt=cnt, C3=C3):
                if y.g is None:
                    return
                g3 = y.g[:, :C3]
                if xl.F.g is None:
                    xl.F.g = torch.empty_like(xl.F.d)
                    E.copy_cols(xl.F.g, 0, g3)
                else:
                    E.add_into(xl.F.g, g3)
                batch_point_sample_level_bwd(xl.cs, V, y.g, C3, pix, cnt, f2d, Hf, Wf)
            E.TAPE.add(bwd)
            outs.append(SparseTensor(xl.cs, y))
        E.mark('A8+A9 projection fusion')
        return outs

    # ------------------------------------------------------------------ reference protocol
    def loss(self, batch_inputs_dict, batch_data_samples, **kwargs):
        x = self.extract_feat(batch_inputs_dict, batch_data_samples)
        return self.bbox_head.loss(x, batch_data_samples, **kwargs)

    def predict(self, batch_inputs_dict, batch_data_samples, **kwargs):
        """sparse_featfusion_single_stage.py:245-280: detections are attached to the data samples as
        `pred_instances_3d` (and returned)."""
        was = self.training
        self.train(False)
        prev = E.TAPE.enabled
        E.TAPE.enabled = False
        try:
            x = self.extract_feat(batch_inputs_dict, batch_data_samples)
            results = self.bbox_head.predict(x, batch_data_samples, **kwargs)
        finally:
            E.TAPE.enabled = prev
            self.train(was)
        for ds, r in zip(batch_data_samples, results):
            ds.pred_instances_3d = r
        return batch_data_samples

    def forward(self, inputs, data_samples=None, mode='tensor', **kwargs):
        hip.refresh_stream()                     # launches follow torch's CURRENT stream of this call
        if mode == 'loss':
            return self.loss(inputs, data_samples, **kwargs)
        elif mode == 'predict':
            return self.predict(inputs, data_samples, **kwargs)
        raise RuntimeError(f'Invalid mode "{mode}". Only supports loss, predict and tensor mode')

    __call__ = forward

    def train_step(self, data, optim_wrapper):
        """mmengine BaseModel.train_step: preprocess, loss forward, sum of the 'loss' entries, backward, update."""
        E.settle_gc(self)                        # (the collector's generations are frozen once the warm-up steps are through)
        E.TAPE.clear()
        hip.refresh_stream()
        E.mark('A1-A3 depth->points')            # whatever the caller queued before train_step (pipeline.make_batch)
        pf = self._pf_take(data)
        if pf is not None:
            data = pf['pre']                     # preprocessed, voxelised and mapped by prefetch() under the previous step
        elif self.data_preprocessor is not None:
            data = self.data_preprocessor(data, True)
        self._bind()
        self.arena.grad.zero_()
        E.new_grad_epoch()                       # first weight-gradient launch per weight overwrites, later ones add
        losses = self.forward(data['inputs'], data['data_samples'], mode='loss')
        E.mark('A10-A16 head fwd + targets + losses')
        if is_dist():
            # bucketed gradient all-reduce overlapped with backward: markers fire when the tape (run in reverse) has
            # finished the head (+ fusion) closures, then the 3-D backbone's, then everything
            if getattr(self.arena, 'reducer', None) is None:
                self.arena.reducer = BucketedGradReducer(self.arena, groups=self._bucket_groups)
            red = self.arena.reducer
            self._backward(red)
        else:
            self._backward(None)
        E.mark('backward (head, 3-D, 2-D)')
        optim_wrapper.update_params(self.arena)
        E.mark('all-reduce wait + clip + AdamW')
        self._pf_done()
        return losses

    # gradient buckets of the data-parallel exchange (parallel.BucketedGradReducer): parts 0 / 1 = the backbones, the
    # implicit last part = everything else (head); subclasses add parts and tape marks for what sits behind the fusion
    _bucket_groups = (('backbone.',), ('backbone_3d.',))

    def _backward(self, red):
        """Run the tape in reverse: head + fusion, then the 3-D backbone on the main stream while the 2-D backbone's
        backward runs on the side stream (they share nothing but the fusion gradients).  `red` (data parallel only)
        all-reduces each part of the gradient arena as soon as it is complete.  `_tape_marks` = [end of the 2-D
        backbone's closures, end of the 3-D backbone's, *(tape index, part) pairs a subclass recorded behind them*]: when
        the reverse replay has passed a pair's index, that part's gradients are complete."""
        fns = E.TAPE.fns
        m2d, m3d = self._tape_marks[:2]
        extra = sorted(self._tape_marks[2:], reverse=True) + [(m3d, 2)]     # part 2: what sits between the 3-D backbone and the marks
        hi = len(fns)
        for idx, part in extra:                                # (the last pair: everything behind the backbones)
            for fn in reversed(fns[idx:hi]):
                fn()
            hi = idx
            if red is not None:
                E.join_wgrad_streams(final=False)
                red.launch(part)                               # e.g. head gradients complete
        img, pts = fns[:m2d][::-1], fns[m2d:m3d][::-1]
        if E.TWO_STREAMS[0] and img and pts:
            # issue the two branches in alternating chunks so that neither queue runs dry while the host is busy
            nchunk = 6
            for c in range(nchunk):
                with E.side_stream(fork=(c == 0)):
                    for fn in img[len(img) * c // nchunk: len(img) * (c + 1) // nchunk]:
                        fn()
                for fn in pts[len(pts) * c // nchunk: len(pts) * (c + 1) // nchunk]:
                    fn()
            if red is not None:
                E.join_wgrad_streams(final=False)
                red.launch(1)                                  # 3-D backbone gradients complete
            E.join_side()
        else:
            for fn in pts:
                fn()
            if red is not None:
                E.join_wgrad_streams(final=False)
                red.launch(1)
            for fn in img:
                fn()
        E.join_wgrad_streams()
        if red is not None:
            red.launch(0)                                      # 2-D backbone gradients complete
        E.TAPE.clear()
