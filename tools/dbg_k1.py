import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from embodiedscan_amd.hip import P, call
dev = torch.device('cuda:0'); st = torch.cuda.current_stream().cuda_stream
rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
for n, cin, cout in ((36352, 320, 128), (36352, 128, 320), (4544, 320, 128), (71, 320, 128), (36352, 256, 128), (36352, 384, 128), (36352, 320, 64)):
    g = torch.Generator().manual_seed(0)
    x = torch.randn(n, cin, generator=g).to(dev); w = (torch.randn(1, cin, cout, generator=g) * 0.05).to(dev)
    wt = torch.empty((1, cout, cin), dtype=torch.bfloat16, device=dev); wn = torch.empty((1, cin, cout), dtype=torch.bfloat16, device=dev)
    call('es_cast_weight_bf16', P(w), 1, cin, cout, P(wn), P(wt), st)
    y = torch.empty(n, cout, device=dev)
    call('es_spconv_fwd_bf16', P(x), 0, cin, P(wt), 0, n, n, 1, cin, cout, 0, P(y), cout, 0, st)
    ref = x @ w[0]
    e1 = rel(y, ref)
    # via the table cast
    tab = torch.tensor([[w.data_ptr(), wn.data_ptr(), wt.data_ptr(), 1, cin, cout, 0]], dtype=torch.int64).to(dev)
    wn.zero_(); wt.zero_()
    call('es_cast_weights_table', P(tab), 1, ((cin + 63) // 64) * ((cout + 63) // 64), st)
    call('es_spconv_fwd_bf16', P(x), 0, cin, P(wt), 0, n, n, 1, cin, cout, 0, P(y), cout, 0, st)
    e2 = rel(y, ref)
    e3 = rel(wt.float(), w.transpose(1, 2)); e4 = rel(wn.float(), w)
    print(f'n={n} {cin}->{cout}: fwd err single-cast {e1:.2e}, table-cast {e2:.2e}; table copies: transposed {e3:.2e} natural {e4:.2e}')
