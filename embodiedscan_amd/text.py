"""Text side of the grounder (embodiedscan/models/detectors/sparse_featfusion_grounder.py:104-116,449-510,570-630).

The reference tokenises with RobertaTokenizerFast('roberta-base') and encodes with a FROZEN RobertaModel
(lr_mult 0.0, configs/grounding/...py:197-199).  Neither the vocabulary nor the weights exist offline, so:
  * `HashTokenizer` is a deterministic stand-in with the slice of the tokenizer protocol the grounder uses (batch encode
    with padding='longest', attention mask, char_to_token) -- `<s> word word ... </s>`, ids from a CRC of the word;
    a real `RobertaTokenizerFast` instance can be passed to the detector instead;
  * the text encoder is transformers' RobertaModel with RANDOM weights (a frozen, external torch module run under
    no_grad: the one place where torch computes on this path; its output is an INPUT of the trainable `text_feat_map`)."""
import re
import zlib
import torch


class _Encoded:
    def __init__(self, input_ids, attention_mask, spans):
        self.input_ids, self.attention_mask, self._spans = input_ids, attention_mask, spans

    def to(self, device):
        self.input_ids, self.attention_mask = self.input_ids.to(device), self.attention_mask.to(device)
        return self

    def char_to_token(self, batch_idx, char_idx):
        for t, (a, b) in enumerate(self._spans[batch_idx]):
            if a <= char_idx < b:
                return t
        return None


class HashTokenizer:
    bos, pad, eos = 0, 1, 2

    def __init__(self, vocab_size=50265):
        self.vocab_size = vocab_size

    def batch_encode_plus(self, texts, padding='longest', return_tensors='pt'):
        ids, spans = [], []
        for t in texts:
            row, sp = [self.bos], [(-1, -1)]
            for m in re.finditer(r'\S+', t):
                row.append(3 + zlib.crc32(m.group(0).lower().encode()) % (self.vocab_size - 3))
                sp.append((m.start(), m.end()))
            row.append(self.eos)
            sp.append((-1, -1))
            ids.append(row)
            spans.append(sp)
        T = max(len(r) for r in ids)
        input_ids = torch.full((len(ids), T), self.pad, dtype=torch.long)
        mask = torch.zeros((len(ids), T), dtype=torch.long)
        for i, r in enumerate(ids):
            input_ids[i, :len(r)] = torch.tensor(r)
            mask[i, :len(r)] = 1
        return _Encoded(input_ids, mask, spans)


def build_text_encoder(cfg=None, seed=0):
    """RobertaModel(RobertaConfig(**cfg)) with random weights, eval mode, no gradients"""
    from transformers import RobertaConfig, RobertaModel
    with torch.random.fork_rng():
        torch.manual_seed(seed)
        model = RobertaModel(RobertaConfig(**(cfg or {})), add_pooling_layer=False)
    model.eval()
    for p in model.parameters():
        p.requires_grad_(False)
    return model


class TextGraph:
    """The frozen encoder's forward for one (B, T) token shape as a captured HIP graph (round 6).  In eager mode the module issues ~ 300
    small launches per call from Python (4.6 ms of a 52 ms grounding step, most of it launch gaps); the weights are frozen and the module
    is in eval mode, so the launch sequence of a shape never changes: captured once on the text stream after a warm-up call, replayed
    afterwards with the token ids / mask copied into the graph's input buffers.  run() must be called with the capture stream current;
    it returns the graph's OUTPUT BUFFER (overwritten by the next replay: the caller copies what it keeps, on the same stream)."""

    def __init__(self, encoder, B, T, device, stream):
        self.ids = torch.ones((B, T), dtype=torch.long, device=device)
        self.mask = torch.ones((B, T), dtype=torch.long, device=device)
        with torch.cuda.stream(stream), torch.no_grad():
            for _ in range(2):                  # warm-up: library handles / workspaces are created outside the capture
                encoder(input_ids=self.ids, attention_mask=self.mask)
        stream.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        # (thread-local capture errors: a loader / feeder thread pinning memory while the capture runs must not invalidate it)
        with torch.no_grad(), torch.cuda.graph(self.graph, stream=stream, capture_error_mode='thread_local'):
            self.out = encoder(input_ids=self.ids, attention_mask=self.mask).last_hidden_state

    def run(self, ids, mask):
        self.ids.copy_(ids, non_blocking=True)
        self.mask.copy_(mask, non_blocking=True)
        self.graph.replay()
        return self.out


def create_positive_map(tokenized, tokens_positive, batch_idx, max_num_entities=256):
    """sparse_featfusion_grounder.py:570-621"""
    positive_map = torch.zeros((len(tokens_positive), max_num_entities), dtype=torch.float)
    for j, tok_list in enumerate(tokens_positive):
        for (beg, end) in tok_list:
            beg_pos = tokenized.char_to_token(batch_idx, beg)
            end_pos = tokenized.char_to_token(batch_idx, end - 1)
            if beg_pos is None:
                beg_pos = tokenized.char_to_token(batch_idx, beg + 1)
                if beg_pos is None:
                    beg_pos = tokenized.char_to_token(batch_idx, beg + 2)
            if end_pos is None:
                end_pos = tokenized.char_to_token(batch_idx, end - 2)
                if end_pos is None:
                    end_pos = tokenized.char_to_token(batch_idx, end - 3)
            if beg_pos is None or end_pos is None:
                continue
            positive_map[j, beg_pos:end_pos + 1].fill_(1)
    return positive_map / (positive_map.sum(-1)[:, None] + 1e-6)
