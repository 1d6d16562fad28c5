"""EXPERIMENTAL (not yet run on a GPU; skipped unless ES_TEST_EXPERIMENTAL=1): the generative transposed convolution of the head's
up-blocks as ONE launch per direction (engine.GEN_FUSED / es_gen_transpose_fwd_bf16 / es_gen_transpose_dgrad_bf16) against the
eight per-tap launches: forward bit-identical (same products, same order), data gradient equal to 1e-6 relative (the taps are
summed in one accumulator chain instead of eight read-modify-write passes), weight gradients untouched."""
import os

import pytest
import torch

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(os.environ.get('ES_TEST_EXPERIMENTAL') != '1', reason='experimental path: set ES_TEST_EXPERIMENTAL=1')]


@pytest.mark.parametrize('n,cin,cout', [(740, 1024, 512), (5920, 512, 256), (47360, 256, 128), (333, 64, 96)])
def test_fused_generative_transpose_matches_the_per_tap_launches(n, cin, cout):
    from embodiedscan_amd import engine as E
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(n)
    x = torch.randn(n, cin, generator=g).to(dev)
    w = (torch.randn(8, cin, cout, generator=g) / cin ** 0.5).to(dev)
    dy = torch.randn(n * 8, cout, generator=g).to(dev)
    res = {}
    E.PRECISION[0] = 'bf16'
    try:
        for fused in (False, True):
            E.GEN_FUSED[0] = fused
            E.TAPE.clear()
            E.WEIGHT_VERSION[0] += 1
            xv, wp = E.Var(x.clone()), E.Param(w.clone(), torch.zeros_like(w))
            y = E.gen_conv_transpose(xv, wp)
            y.g = dy.clone()
            E.TAPE.backward()
            torch.cuda.synchronize()
            res[fused] = (y.d.clone(), xv.g.clone(), wp.g.clone())
    finally:
        E.GEN_FUSED[0] = False
        E.PRECISION[0] = 'f32'
    assert torch.equal(res[True][0], res[False][0]), float((res[True][0] - res[False][0]).abs().max())
    e = float((res[True][1] - res[False][1]).norm() / res[False][1].norm())
    assert e < 1e-6, e
    assert torch.equal(res[True][2], res[False][2])
    print(f'fused generative transpose n={n} {cin}->{cout}: forward identical, data gradient rel-L2 {e:.1e}, weight gradient identical')
