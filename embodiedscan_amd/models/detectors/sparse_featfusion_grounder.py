"""SparseFeatureFusion3DGrounder (embodiedscan/models/detectors/sparse_featfusion_grounder.py:29-766) on the MI355X
kernels.  Same registry name, constructor arguments and forward(inputs, data_samples, mode) protocol.

extract_feat is the mv-3ddet feature path (2-D / 3-D backbones + projection fusion, inherited) followed by MinkNeck;
pre_decoder / forward_decoder / the head run on padded channels-last token matrices.  Text: see embodiedscan_amd/text.py."""
import os
import torch
from ... import engine as E
from ... import hip
from ...hip import P, call
from ...params import ParamArena, grounder_specs
from ...registry import MODELS
from ...text import HashTokenizer, TextGraph, build_text_encoder, create_positive_map
from ..layers.ground_transformer.decoder import SparseFeatureFusionTransformerDecoder, _Lin
from .sparse_featfusion_single_stage import SparseFeatureFusionSingleStage3DDetector

TEXT_GRAPH = [os.environ.get('ES_TEXT_GRAPH', '1') != '0']     # ... as a captured graph per token shape (text.TextGraph)
TEXT_ASYNC = [os.environ.get('ES_TEXT_ASYNC', '1') != '0']     # round 6: the frozen text encoder on its own stream, queued before the backbones


@MODELS.register_module()
class SparseFeatureFusion3DGrounder(SparseFeatureFusionSingleStage3DDetector):
    _version = 2

    def __init__(self, backbone, backbone_3d, bbox_head, neck=None, neck_3d=None, decoder=None, voxel_size=0.01,
                 num_queries=512, max_num_entities=256, coord_type='CAMERA', train_cfg=None, test_cfg=None,
                 data_preprocessor=None, use_xyz_feat=False, init_cfg=None, seed=0, device='cuda:0', text_encoder_cfg=None,
                 tokenizer=None):
        assert neck is None and neck_3d is not None and decoder is not None
        self.device = torch.device(device)
        self.backbone = MODELS.build(backbone)
        self.backbone.act16 = True              # feature maps only feed the projection fusion: bf16 activation storage
        self.backbone_3d = MODELS.build(backbone_3d)
        self.neck_3d = MODELS.build(neck_3d)
        bbox_head = dict(bbox_head)
        bbox_head.update(train_cfg=train_cfg, test_cfg=test_cfg)
        self.bbox_head = MODELS.build(bbox_head)
        self.decoder = SparseFeatureFusionTransformerDecoder(**decoder)
        self.embed_dims = self.decoder.embed_dims
        self.data_preprocessor = MODELS.build(data_preprocessor, device=self.device) if data_preprocessor else None
        self.coord_type, self.train_cfg, self.test_cfg = coord_type, train_cfg, test_cfg
        self.num_queries = num_queries
        self.max_num_entities = self.bbox_head.contrastive_cfg.get('max_text_len', max_num_entities)
        self.voxel_size, self.use_xyz_feat = voxel_size, use_xyz_feat
        # text side (roberta-base vocabulary / weights are not available offline: stand-ins, see text.py)
        self.tokenizer = tokenizer or HashTokenizer()
        self.text_encoder = build_text_encoder(text_encoder_cfg, seed=seed)
        self.text_dim = self.text_encoder.config.hidden_size
        self.arena = ParamArena(grounder_specs(text_dim=self.text_dim, E=self.embed_dims, num_layers=self.decoder.num_layers,
                                               ffn=self.decoder.ffn_channels, in_channels=self.neck_3d.in_channels), seed=seed)
        self.training = True
        self._bound = False
        self._pf_init()

    # gradient buckets (data parallel): the backbones, the MinkNeck, and -- implicit last part -- decoder + head +
    # text_feat_map, whose gradients are complete first: four all-reduces, each under the backward of what precedes it
    _bucket_groups = (('backbone.',), ('backbone_3d.',), ('neck_3d.',))

    # ------------------------------------------------------------------ parameters
    def to(self, device):
        super().to(device)
        self.text_encoder.to(self.device)
        return self

    def _bind(self):
        if not self._bound:
            if self.arena.data.device != self.device:
                self.arena.to(self.device)
            if next(self.text_encoder.parameters()).device != self.device:
                self.text_encoder.to(self.device)
            E.begin_bind(id(self))
            try:
                self.backbone.bind(self.arena, 'backbone.')
                self.backbone_3d.bind(self.arena, 'backbone_3d.')
                self.neck_3d.bind(self.arena, 'neck_3d.')
                self.decoder.bind(self.arena, 'decoder.')
                self.bbox_head.bind(self.arena, 'bbox_head.')
                self.text_feat_map = _Lin(self.arena, 'text_feat_map.weight', 'text_feat_map.bias')
            finally:
                E.end_bind()
            self._bound = True

    # The frozen RoBERTa is a submodule of the reference detector, so its weights are part of the reference's state dict
    # (`text_encoder.*`, sparse_featfusion_grounder.py:104-116): a checkpoint written here must carry them and a reference
    # grounding checkpoint must load them -- otherwise the prompts would be encoded by random weights (round-2 advisor).
    def state_dict(self):
        sd = self.arena.state_dict()
        for k, v in self.text_encoder.state_dict().items():
            sd['text_encoder.' + k] = v
        return sd

    def load_state_dict(self, sd, strict=False, tap_order=None):
        pre = 'text_encoder.'
        text = {k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)}
        rest = {k: v for k, v in sd.items() if not k.startswith(pre)}
        missing, unexpected = super().load_state_dict(rest, strict=False, tap_order=tap_order)
        own = self.text_encoder.state_dict()
        dev = next(self.text_encoder.parameters()).device
        with torch.no_grad():
            for k, v in text.items():
                if k in own and tuple(own[k].shape) == tuple(v.shape):
                    own[k].copy_(v.to(dev))
                else:
                    unexpected.append(pre + k)
        # (buffers such as embeddings.position_ids exist or not depending on the transformers version: never "missing")
        params = {k for k, _ in self.text_encoder.named_parameters()}
        missing = list(missing) + [pre + k for k in own if k not in text and k in params]
        if strict and (missing or unexpected):
            raise RuntimeError(f'load_state_dict: missing {missing[:5]}... unexpected {unexpected[:5]}...')
        return missing, unexpected

    def train(self, mode=True):
        self.training = mode
        for m in (self.backbone_3d, self.neck_3d, self.decoder, self.bbox_head):
            m.training = mode
        return self

    # ------------------------------------------------------------------ text
    def start_text(self, batch_data_samples):
        """:475-481, first half of encode_text: tokenise, positive maps, frozen RoBERTa.  The encoder depends on the prompts only, so it is
        queued on its OWN stream before the backbones (round 6: its ~ 300 small launches used to sit between the neck and the decoder
        on the main stream, 4.6 ms of a 52.6 ms step); finish_text() joins.  ES_TEXT_ASYNC=0: on the calling stream, as before."""
        texts = [ds.text for ds in batch_data_samples]
        tok = self.tokenizer.batch_encode_plus(texts, padding='longest', return_tensors='pt')
        if all(getattr(ds, 'tokens_positive', None) is not None for ds in batch_data_samples):
            tps = [ds.tokens_positive for ds in batch_data_samples]
        else:
            tps = [[[0, 1]] for _ in batch_data_samples]
        pmaps = [create_positive_map(tok, tp, i, self.max_num_entities) for i, tp in enumerate(tps)]
        side = TEXT_ASYNC[0] and self.device.type == 'cuda'
        ev = None
        if side:
            if getattr(self, '_text_stream', None) is None:
                # the weight-gradient stream of the compute stream (idle during the forward pass) unless told otherwise: a stream of its
                # own is a fifth .. seventh stream on four hardware queues (ES_TEXT_STREAM=own: A/B)
                own = os.environ.get('ES_TEXT_STREAM', 'wgrad') == 'own' or not E.WGRAD_ASYNC[0]
                self._text_stream = torch.cuda.Stream(device=self.device) if own else E.wgrad_stream_obj()
            ctx = torch.cuda.stream(self._text_stream)
        else:
            import contextlib
            ctx = contextlib.nullcontext()
        with ctx:
            tok = tok.to(self.device)
            with torch.no_grad():
                B, T = tok.input_ids.shape
                hs = self._encode_graph(tok, B, T) if (side and TEXT_GRAPH[0]) else None
                if hs is None:
                    hs = self.text_encoder(input_ids=tok.input_ids, attention_mask=tok.attention_mask).last_hidden_state
                hs32 = hs.reshape(B * T, self.text_dim).float().contiguous()
                tlen = tok.attention_mask.sum(1).to(torch.int32).contiguous()
                mask = tok.attention_mask.bool()
            if side:
                ev = torch.cuda.Event()
                ev.record(self._text_stream)
        return dict(tok=tok, pmaps=pmaps, hs=hs, hs32=hs32, tlen=tlen, mask=mask, ev=ev, B=B, T=T)

    def _encode_graph(self, tok, B, T):
        """the frozen encoder as a graph replay per (B, T) token shape (text.TextGraph); None: this shape runs eagerly (capture failed once,
        or more than 16 shapes are alive).  Called with the text stream current."""
        graphs = self.__dict__.setdefault('_text_graphs', {})
        g = graphs.get((B, T))
        if g is None and len(graphs) < 16:
            try:
                g = TextGraph(self.text_encoder, B, T, self.device, self._text_stream)
            except Exception as exc:            # (a capture the libraries underneath do not allow: keep the eager path, say so once)
                import warnings
                warnings.warn(f'text encoder graph capture failed for shape {(B, T)}: {exc!r}; running eagerly')
                torch.cuda.synchronize(self.device)
                g = False
            graphs[(B, T)] = g
        if not g:
            return None
        return g.run(tok.input_ids, tok.attention_mask).clone()

    def finish_text(self, job, batch_data_samples):
        """:482-498, second half: the calling stream waits for the encoder, then text_feat_map (trainable, recorded on the tape HERE)"""
        if job['ev'] is not None:
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(job['ev'])
            for t in (job['hs'], job['hs32'], job['tlen'], job['mask'], job['tok'].input_ids, job['tok'].attention_mask):
                t.record_stream(cur)             # allocated on the text stream, read on this one
        mask, B, T = job['mask'], job['B'], job['T']
        for i, ds in enumerate(batch_data_samples):
            pm = job['pmaps'][i].bool().float()
            ds.gt_instances_3d.positive_maps = pm
            ds.gt_instances_3d.text_token_mask = mask[i].unsqueeze(0).repeat(len(pm), 1)
        text = self.text_feat_map(E.Var(job['hs32'], rg=False), need_dx=False)
        self.last_text = dict(hidden=job['hs'], mask=mask, input_ids=job['tok'].input_ids)
        return text, mask, job['tlen'], T

    def encode_text(self, batch_data_samples):
        """:475-498: tokenise, positive maps, frozen RoBERTa, text_feat_map.  Returns (text Var (B*T, E), mask (B,T) bool
        on the device, tlen (B,) int32 on the device, T) and attaches positive_maps / text_token_mask to the samples."""
        return self.finish_text(self.start_text(batch_data_samples), batch_data_samples)

    # ------------------------------------------------------------------ features
    def extract_feat(self, batch_inputs_dict, batch_data_samples):
        """:176-310: fused sparse levels (inherited) -> MinkNeck -> (feats, scores, coords) per-sample lists"""
        x = super().extract_feat(batch_inputs_dict, batch_data_samples)
        out = self.neck_3d(x, len(batch_data_samples))
        # everything recorded from here on (text map, decoder, head) belongs to the last gradient part (3); once the reverse
        # replay is back here only fusion + neck closures remain before the 3-D backbone's: part 2 (neck_3d.) is theirs
        self._tape_marks = self._tape_marks[:2] + [(len(E.TAPE.fns), 3)]
        return out

    def forward_transformer(self, text, tlen, T, batch_data_samples):
        """pre_decoder + forward_decoder (:312-447) on the padded buffers of the neck"""
        nk = self.neck_3d.last
        feats, coords, lens, Lmax = nk['feats'], nk['points'], nk['lens'], nk['Lmax']
        B = len(lens)
        dev = feats.d.device
        s = hip.stream()
        klen = torch.tensor(lens, dtype=torch.int32).to(dev, non_blocking=True)
        # query selection: ContrastiveEmbed scores of every point token, max over text tokens, top-k per sample
        _, rowmax = self.bbox_head.cls_branch(feats, text, B, Lmax, T, tlen, vlen=klen, want_logits=False, want_max=True)
        Q = min(self.num_queries, min(lens))
        idx = torch.empty((B, Q), dtype=torch.int32, device=dev)
        call('es_topk_sorted', P(rowmax), B, Lmax, P(klen), Q, P(idx), s)
        if getattr(self, 'force_queries', None) is not None:
            # test hook (teacher forcing, tests/test_gpu_grounding.py): take the oracle's query indices so that a bf16 run can be
            # compared with its specification element by element even when the top-k boundary would flip a near tie
            self.free_queries = idx
            idx = self.force_queries.to(device=dev, dtype=torch.int32).reshape(B, Q).contiguous()
        gidx = (idx + (torch.arange(B, device=dev, dtype=torch.int32) * Lmax)[:, None]).reshape(-1).contiguous()
        query = E.gather_rows(feats, gidx)
        qcoords = torch.empty((B * Q, 3), dtype=torch.float32, device=dev)
        call('es_row_move', P(qcoords), 3, P(coords), 3, P(gidx), B * Q, 3, 0, s)
        prev = E.TAPE.enabled
        E.TAPE.enabled = False                   # proposals: reg_branches[num_layers] on the selected tokens, detached
        try:
            pred0 = self.bbox_head.decode(qcoords, self.bbox_head.reg_branch(E.Var(query.d, rg=False))).d
        finally:
            E.TAPE.enabled = prev
        self.last_queries = dict(idx=idx, gidx=gidx, rowmax=rowmax, pred0=pred0, Q=Q, klen=klen)
        return self.decoder(query, feats, coords, qcoords, pred0, text, B, Q, Lmax, T, klen, tlen, self.bbox_head)

    # ------------------------------------------------------------------ reference protocol
    def loss(self, batch_inputs_dict, batch_data_samples, **kwargs):
        self._bind()
        job = self.start_text(batch_data_samples)
        self.extract_feat(batch_inputs_dict, batch_data_samples)
        E.mark('A19 MinkNeck (+ pruning, token padding)')
        text, mask, tlen, T = self.finish_text(job, batch_data_samples)
        E.mark('A19 frozen text encoder + text_feat_map')
        hidden, boxes = self.forward_transformer(text, tlen, T, batch_data_samples)
        E.mark('A19 query selection + 6-layer decoder')
        out = self.bbox_head.loss(hidden, boxes, text, mask, batch_data_samples, tlen=tlen)
        E.mark('A19/N2 head: token logits + Hungarian + focal + corner-Chamfer')
        return out

    def predict(self, batch_inputs_dict, batch_data_samples, **kwargs):
        was = self.training
        self.train(False)
        prev = E.TAPE.enabled
        E.TAPE.enabled = False
        try:
            self._bind()
            job = self.start_text(batch_data_samples)
            self.extract_feat(batch_inputs_dict, batch_data_samples)
            text, mask, tlen, T = self.finish_text(job, batch_data_samples)
            hidden, boxes = self.forward_transformer(text, tlen, T, batch_data_samples)
            results = self.bbox_head.predict(hidden, boxes, text, mask, batch_data_samples, tlen=tlen)
        finally:
            E.TAPE.enabled = prev
            self.train(was)
        for ds, r in zip(batch_data_samples, results):
            ds.pred_instances_3d = r
        return batch_data_samples
