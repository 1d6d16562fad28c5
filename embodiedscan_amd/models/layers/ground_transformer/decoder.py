"""SparseFeatureFusionTransformerDecoder (embodiedscan/models/layers/ground_transformer/decoder.py:20-297) on the MI355X
kernels.  Tokens are channels-last row matrices: queries (B*Q, E), padded point tokens (B*Lmax, E), padded text tokens
(B*T, E).  Projections and the FFN are row GEMMs on the convolution engine (bf16 MFMA in bf16 mode), the attention cores
are the flash-style MFMA kernels of csrc/transformer.hip, LayerNorm (+ residual) and the learned position embeddings
(Conv1d - BatchNorm1d (train mode: batch statistics over ALL B*L rows, padded ones included, as in the reference) - ReLU -
Conv1d) are fused row kernels.  mmcv's wrappers are restated: MultiheadAttention(batch_first) = identity + attn(q + q_pos,
k + k_pos, v) (value gets no position), FFN = x + Linear(ReLU(Linear(x))); dropout 0."""
import torch
from .... import engine as E


class _Lin:
    def __init__(self, arena, name_w, name_b):
        self.w = E.Param(arena.p[name_w], arena.g.get(name_w))
        self.b = E.Param(arena.p[name_b], arena.g.get(name_b)) if name_b else None

    def __call__(self, x, need_dx=True):
        return E.linear(x, self.w, self.b, need_dx=need_dx)


class _LN:
    def __init__(self, arena, p):
        self.w = E.Param(arena.p[p + '.weight'], arena.g.get(p + '.weight'))
        self.b = E.Param(arena.p[p + '.bias'], arena.g.get(p + '.bias'))

    def __call__(self, x, res=None):
        return E.layernorm(x, self.w, self.b, res=res)


class PositionEmbeddingLearned:
    """decoder.py:20-34.  `repeat`: how many times the reference evaluates it on the same input within one step (the
    cross_posembed of the key coordinates is recomputed in each of the 6 layers): the forward runs once, the BatchNorm
    running statistics receive the equivalent momentum 1 - (1 - 0.1)**repeat."""

    def __init__(self, arena, p):
        q = p + '.position_embedding_head'
        self.l0 = _Lin(arena, q + '.0.weight', q + '.0.bias')
        self.bn_w = E.Param(arena.p[q + '.1.weight'], arena.g.get(q + '.1.weight'))
        self.bn_b = E.Param(arena.p[q + '.1.bias'], arena.g.get(q + '.1.bias'))
        self.running = (arena.p[q + '.1.running_mean'], arena.p[q + '.1.running_var'])
        self.l3 = _Lin(arena, q + '.3.weight', q + '.3.bias')

    def __call__(self, xyz, training=True, repeat=1):
        """xyz: raw (n, c) device tensor (no gradient: box / coordinate inputs are detached in the reference)"""
        h = self.l0(E.Var(xyz, rg=False), need_dx=False)
        n = h.d.shape[0]
        if training:
            h = E.norm(h, self.bn_w, self.bn_b, [0, n], 1e-5, act=1, running=self.running, momentum=1.0 - 0.9 ** repeat)
        else:                                   # eval: running statistics folded to scale / shift, + ReLU
            from .... import hip
            from ....hip import P, call
            C = self.bn_w.d.numel()
            sc, sh = torch.empty(C, dtype=torch.float32, device=xyz.device), torch.empty(C, dtype=torch.float32, device=xyz.device)
            call('es_bn_fold', P(self.bn_w.d), P(self.bn_b.d), P(self.running[0]), P(self.running[1]), C, 1e-5, P(sc), P(sh), hip.stream())
            y = E.Var(torch.empty_like(h.d), rg=False)
            call('es_affine_act_fwd', P(h.d), P(sc), P(sh), 0, n, C, 1, P(y.d), hip.stream())
            h = y
        return self.l3(h)


class _MHA:
    """mmcv MultiheadAttention wrapping nn.MultiheadAttention (embed E, H heads, batch_first, dropout 0)"""

    def __init__(self, arena, p, H):
        w = E.Param(arena.p[p + '.attn.in_proj_weight'], arena.g.get(p + '.attn.in_proj_weight'))
        self.wq, self.wk, self.wv = E.ParamSlice(w, 0), E.ParamSlice(w, 1), E.ParamSlice(w, 2)
        self.in_b = arena.p[p + '.attn.in_proj_bias']
        self.in_bg = arena.g.get(p + '.attn.in_proj_bias')
        self.bias = [E.Param(self.in_b[j], self.in_bg[j] if self.in_bg is not None else None) for j in range(3)]
        self.out = _Lin(arena, p + '.attn.out_proj.weight', p + '.attn.out_proj.bias')
        self.H = H

    def __call__(self, query, q_in, k_in, v_in, B, Lq, Lk, klen):
        """identity (= query) + out_proj(attention(q_in Wq, k_in Wk, v_in Wv))"""
        q = E.linear(q_in, self.wq, self.bias[0])
        k = E.linear(k_in, self.wk, self.bias[1])
        v = E.linear(v_in, self.wv, self.bias[2])
        o = E.attention(q, k, v, B, self.H, Lq, Lk, klen)
        return self.out(o), query        # (attention branch, identity): the caller fuses the add into the LayerNorm


class DecoderLayer:
    def __init__(self, arena, p, H):
        self.self_attn = _MHA(arena, p + 'self_attn', H)
        self.cross_attn_text = _MHA(arena, p + 'cross_attn_text', H)
        self.cross_attn = _MHA(arena, p + 'cross_attn', H)
        self.ffn0 = _Lin(arena, p + 'ffn.layers.0.0.weight', p + 'ffn.layers.0.0.bias')
        self.ffn1 = _Lin(arena, p + 'ffn.layers.1.weight', p + 'ffn.layers.1.bias')
        self.norms = [_LN(arena, p + f'norms.{k}') for k in range(4)]

    def __call__(self, query, query_pos, key, key_with_pos, text, B, Q, Lk, T, klen, tlen):
        """decoder.py:103-179"""
        qp = E.add(query, query_pos)
        a, idt = self.self_attn(query, qp, qp, query, B, Q, Q, None)
        query = self.norms[0](a, res=idt)
        qp = E.add(query, query_pos)
        a, idt = self.cross_attn_text(query, qp, text, text, B, Q, T, tlen)
        query = self.norms[1](a, res=idt)
        qp = E.add(query, query_pos)
        a, idt = self.cross_attn(query, qp, key_with_pos, key, B, Q, Lk, klen)
        query = self.norms[2](a, res=idt)
        h = E.relu_(self.ffn0(query))
        return self.norms[3](self.ffn1(h), res=query)


class SparseFeatureFusionTransformerDecoder:
    def __init__(self, num_layers, layer_cfg, post_norm_cfg=dict(type='LN'), return_intermediate=True, init_cfg=None):
        if post_norm_cfg is not None:
            raise ValueError('There is not post_norm in SparseFeatureFusionTransformerDecoder')
        self.num_layers, self.layer_cfg, self.return_intermediate = num_layers, layer_cfg, return_intermediate
        sa = layer_cfg.get('self_attn_cfg', {})
        self.embed_dims, self.num_heads = sa.get('embed_dims', 256), sa.get('num_heads', 8)
        assert self.embed_dims // self.num_heads == 32, 'the attention kernels are built for head_dim 32 (256 / 8)'
        self.ffn_channels = layer_cfg.get('ffn_cfg', {}).get('feedforward_channels', 1024)
        self.training = True

    def bind(self, arena, prefix='decoder.'):
        self.layers = [DecoderLayer(arena, f'{prefix}layers.{i}.', self.num_heads) for i in range(self.num_layers)]
        self.self_posembed = PositionEmbeddingLearned(arena, prefix + 'self_posembed')
        self.cross_posembed = PositionEmbeddingLearned(arena, prefix + 'cross_posembed')
        self.norm = _LN(arena, prefix + 'norm')
        return self

    def forward(self, query, key, key_coords, query_coords, pred_bboxes, text, B, Q, Lk, T, klen, tlen, bbox_head):
        """decoder.py:224-297.  query Var (B*Q, E); key Var (B*Lk, E) padded; key_coords (B*Lk, 3), query_coords (B*Q, 3),
        pred_bboxes (B*Q, 9) raw tensors; text Var (B*T, E).  Returns (hidden states [Var] per layer (after the shared
        LayerNorm), predicted boxes [Var (B*Q, 9)] per layer)."""
        tr = self.training
        key_pos = self.cross_posembed(key_coords, tr, repeat=self.num_layers)  # identical in all layers: computed once
        key_with_pos = E.add(key, key_pos)
        inter, boxes = [], []
        for lid, layer in enumerate(self.layers):
            query_pos = self.self_posembed(pred_bboxes, tr)
            query = layer(query, query_pos, key, key_with_pos, text, B, Q, Lk, T, klen, tlen)
            reg = bbox_head.reg_branch(query)
            new_boxes = bbox_head.decode(query_coords, reg)
            pred_bboxes = new_boxes.d                                          # .detach(): the next layer's position input
            inter.append(self.norm(query))
            boxes.append(new_boxes)
        return inter, boxes

    __call__ = forward
