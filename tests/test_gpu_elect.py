"""Stress test of the last-workgroup elections behind the deterministic in-launch reductions (csrc/common.h): thousands of
es_colsum / es_layernorm_bwd launches on one stream while large kernels stream through the L2s of every XCD on two other streams.
A reduction that read a stale partial would differ from the f64 column sums AND from the first launch's bits (the partials are
added in workgroup order: every launch on the same data must return the same bits).  Both election forms (es_set_option key 18:
0 = coherent stores + drained ticket, the default; 1 = agent-scope release / acquire fences) must pass and agree bit for bit."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_last_block_elections_under_load():
    from embodiedscan_amd import hip
    from embodiedscan_amd.hip import call, P
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(9)
    main = torch.cuda.Stream()
    noise = [torch.cuda.Stream(), torch.cuda.Stream()]
    big_a = torch.randn(8192, 8192, device=dev)
    big_b = torch.empty_like(big_a)
    shapes = [(70000, 256), (4100, 512), (300, 128), (20000, 96)]
    data = [torch.randn(n, C, generator=g).to(dev) for n, C in shapes]
    want = [d.double().sum(0) for d in data]
    # LayerNorm backward operands (parameter gradients through the same election)
    n_ln, C_ln = 6000, 256
    dy, z = torch.randn(n_ln, C_ln, generator=g).to(dev), torch.randn(n_ln, C_ln, generator=g).to(dev)
    w = torch.randn(C_ln, generator=g).to(dev)
    mean, var = z.mean(1), z.var(1, unbiased=False)
    rstd = (var + 1e-5).rsqrt()
    xh = (z - mean[:, None]) * rstd[:, None]
    want_dw, want_db = (dy.double() * xh.double()).sum(0), dy.double().sum(0)
    opt = hip.raw('es_set_option')
    results = {}
    try:
        for safe in (0, 1):
            opt(18, safe)
            first = {}
            torch.cuda.synchronize()
            stop = 400 if safe == 0 else 150
            with torch.cuda.stream(main):
                ws = [torch.zeros(int(hip.raw('es_colsum_workspace_floats')(n, C)) + 16, device=dev) for n, C in shapes]
                ws_ln = torch.zeros(int(hip.raw('es_layernorm_bwd_workspace_floats')(n_ln, C_ln)) + 16, device=dev)
                outs = []
                for it in range(stop):
                    for ns in noise:                      # large streaming kernels on the other streams (dirty lines in every L2)
                        with torch.cuda.stream(ns):
                            torch.add(big_a, 1.0, out=big_b)
                    for i, ((n, C), d) in enumerate(zip(shapes, data)):
                        o = torch.empty(C, device=dev)
                        call('es_colsum', P(d), C, n, C, P(o), 0, P(ws[i]), ws[i].numel(), main.cuda_stream)
                        outs.append((i, o))
                    dz = torch.empty_like(dy)
                    dw, db = torch.zeros(C_ln, device=dev), torch.zeros(C_ln, device=dev)
                    call('es_layernorm_bwd', P(dy), P(z), n_ln, C_ln, P(w), P(mean), P(rstd), P(dz), 0, P(dw), P(db), P(ws_ln), ws_ln.numel(),
                         main.cuda_stream)
                    outs.append(('dw', dw))
                    outs.append(('db', db))
            torch.cuda.synchronize()
            for key, o in outs:
                if key not in first:
                    first[key] = o
                    ref = want[key] if isinstance(key, int) else (want_dw if key == 'dw' else want_db)
                    err = float((o.double() - ref).abs().max() / ref.abs().max())
                    assert err < 1e-5, (safe, key, err)
                else:
                    assert torch.equal(o, first[key]), (safe, key, 'a later launch differs from the first: stale partial')
            results[safe] = first
        for key in results[0]:
            assert torch.equal(results[0][key], results[1][key]), key
        print(f'elections under load: {400 * (len(shapes) + 1)} light + {150 * (len(shapes) + 1)} fenced launches, all bit-identical, within 1e-5 of f64')
    finally:
        opt(18, 0)
