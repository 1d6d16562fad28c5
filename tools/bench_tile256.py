"""A/B of the 256-row gather tile written at the end of round 4 (csrc/next/spconv_tile256.hip, built as libes_next.so) against the
shipped LDS-DMA / ping-pong kernels on sparse 27-tap maps of mv-3ddet's sizes -- run once, then promoted or deleted."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def timed(fn, reps=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2]


def main():
    from embodiedscan_amd import hip, sparse
    from embodiedscan_amd.hip import call, P
    nxt = ctypes.CDLL(os.path.join(ROOT, 'embodiedscan_amd', 'libes_next.so'))
    V, I = ctypes.c_void_p, ctypes.c_int
    f = nxt.es_next_spconv_fwd_bf16_tile
    f.restype, f.argtypes = I, [V, I, V, V, I, I, I, I, I, V, V, I, I, V, V, V, I, I, I, I, I, I, V]
    dev = torch.device('cuda:0')
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(1)
    opt = hip.raw('es_set_option')
    for npts, vs in ((400000, 0.01), (100000, 0.02)):
        pts = [(torch.rand(npts, 3, generator=g) * torch.tensor([6.0, 6.0, 2.5])).to(dev)]
        cs, _ = sparse.voxelize(pts, vs)
        nbr = cs.kernel_map(cs, 3)
        n = cs.n
        pairs = float((nbr >= 0).sum())
        for cin, cout in ((64, 64), (128, 128), (256, 256), (64, 128)):
            x = torch.randn(n, cin, generator=g).to(dev).bfloat16().contiguous()
            w = (torch.randn(27, cin, cout, generator=g) / (27 * cin) ** 0.5).to(dev)
            wt = torch.empty((27, cout, cin), dtype=torch.bfloat16, device=dev)
            wn = torch.empty((27, cin, cout), dtype=torch.bfloat16, device=dev)
            call('es_cast_weight_bf16', P(w), 27, cin, cout, P(wn), P(wt), st)
            y0, y1 = torch.empty(n, cout, device=dev), torch.empty(n, cout, device=dev)
            fl = 2 * pairs * cin * cout
            opt(11, 768)
            t0 = timed(lambda: call('es_spconv_fwd_bf16', P(x), 1, cin, P(wt), P(nbr), n, n, 27, cin, cout, 0, P(y0), cout, 0, st))
            opt(11, 0)
            t1 = timed(lambda: call('es_spconv_fwd_bf16', P(x), 1, cin, P(wt), P(nbr), n, n, 27, cin, cout, 0, P(y0), cout, 0, st))
            opt(11, 768)
            row = f'rows {n:7d} {cin:3d}->{cout:3d} pairs/row {pairs / n:5.1f}: ping-pong {t0:7.3f} ms {fl / t0 / 1e9:6.1f} TF | dma128 {t1:7.3f} ms {fl / t1 / 1e9:6.1f} TF'
            for cols, chunk in ((0, 2), (0, 1), (256, 2)):
                if cols == 256 and cout % 256:
                    continue
                rc = f(P(x), cin, P(wt), P(nbr), n, n, 27, cin, cout, 0, P(y1), cout, 0, 0, 0, 0, 0, 0, 0, 256, cols, chunk, st)
                if rc != 0:
                    row += f' | t256 cols={cols} chunk={chunk} rc {rc}'
                    continue
                torch.cuda.synchronize()
                same = bool(torch.equal(y0, y1))
                t2 = timed(lambda: f(P(x), cin, P(wt), P(nbr), n, n, 27, cin, cout, 0, P(y1), cout, 0, 0, 0, 0, 0, 0, 0, 256, cols, chunk, st))
                row += f' | t256 cols={cols} chunk={chunk} {t2:7.3f} ms {fl / t2 / 1e9:6.1f} TF same={same}'
            print(row, flush=True)


if __name__ == '__main__':
    main()
