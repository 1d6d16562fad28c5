"""dev tool: the frozen text encoder (transformers RobertaModel, eval, no grad) eager vs as a captured graph replayed on a side stream --
same outputs? launch counts and times.
  python tools/probe_text_graph.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from embodiedscan_amd.text import build_text_encoder, TextGraph

dev = torch.device('cuda:0')
enc = build_text_encoder(None, seed=0).to(dev)
g = torch.Generator().manual_seed(1)
B, T = 12, 23
ids = torch.randint(3, 50000, (B, T), generator=g).to(dev)
mask = torch.ones((B, T), dtype=torch.long)
for b in range(B):
    mask[b, T - (b % 7):] = 0
mask = mask.to(dev)
ids = torch.where(mask.bool(), ids, torch.ones_like(ids))
side = torch.cuda.Stream()


def eager():
    with torch.no_grad():
        return enc(input_ids=ids, attention_mask=mask).last_hidden_state


ref = eager()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    eager()
torch.cuda.synchronize()
print(f'eager: {(time.perf_counter() - t0) / 10 * 1e3:.2f} ms per call (host + device, one stream)')
tg = TextGraph(enc, B, T, dev, side)
with torch.cuda.stream(side):
    out = tg.run(ids, mask).clone()
torch.cuda.synchronize()
print('graph vs eager: max abs diff', float((out - ref).abs().max()), 'of', float(ref.abs().max()))
t0 = time.perf_counter()
with torch.cuda.stream(side):
    for _ in range(10):
        tg.run(ids, mask)
torch.cuda.synchronize()
print(f'graph replay: {(time.perf_counter() - t0) / 10 * 1e3:.2f} ms per call')
ids2 = torch.randint(3, 50000, (B, T), generator=g).to(dev)
with torch.cuda.stream(side):
    out2 = tg.run(ids2, mask).clone()
with torch.no_grad():
    ref2 = enc(input_ids=ids2, attention_mask=mask).last_hidden_state
torch.cuda.synchronize()
print('second input, graph vs eager: max abs diff', float((out2 - ref2).abs().max()))
