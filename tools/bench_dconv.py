"""A/B of the dense-volume convolution engine (csrc/dconv.hip) against the neighbour-map kernels on the occupancy neck's shapes:
    python tools/bench_dconv.py [--reps 5] > gpurun_out/dconv.txt
per (level, direction): ms and algorithmic TFLOP/s (2 * valid pairs * Cin * Cout) of the map kernel and of the dense kernel under
each tuning variant (row tile 256 / 320, loop order, slices)."""
import argparse
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def timed(fn, reps):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2], ts[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reps', type=int, default=5)
    ap.add_argument('--quick', action='store_true')
    args = ap.parse_args()
    from embodiedscan_amd import hip
    from embodiedscan_amd.hip import call, P
    from embodiedscan_amd.models.necks.imvoxel_neck import VolumeGrid
    dev = torch.device('cuda:0')
    st_ = torch.cuda.current_stream().cuda_stream
    opt = hip.raw('es_dconv_set_option')
    gen = torch.Generator().manual_seed(0)
    shapes = [(40, 40, 16, 768, 768, 1), (20, 20, 8, 1536, 1536, 1), (10, 10, 4, 3072, 3072, 1), (40, 40, 16, 768, 1536, 2),
              (20, 20, 8, 1536, 3072, 2)]
    if args.quick:
        shapes = shapes[:2]
    for X, Y, Z, cin, cout, st in shapes:
        g = (ctypes.c_int * 7)(1, X, Y, Z, 3, st, 1)
        grid = VolumeGrid(1, X, Y, Z, dev)
        nbr, inv, n_out, _ = grid.conv_map(3, st, 1)
        n_in = X * Y * Z
        pairs = float((nbr >= 0).sum())
        flop = 2.0 * pairs * cin * cout
        x = torch.randn(n_in, cin, generator=gen).to(dev)
        w = (torch.randn(27, cin, cout, generator=gen) / (27 * cin) ** 0.5).to(dev)
        xh = x.bfloat16().contiguous()
        wt = torch.empty((27, cout, cin), dtype=torch.bfloat16, device=dev)
        wn = torch.empty((27, cin, cout), dtype=torch.bfloat16, device=dev)
        call('es_cast_weight_bf16', P(w), 27, cin, cout, P(wn), P(wt), st_)
        dy = torch.randn(n_out, cout, generator=gen).to(dev)
        dyh = dy.bfloat16().contiguous()
        y = torch.empty(n_out, cout, device=dev)
        dx = torch.empty(n_in, cin, device=dev)
        dw = torch.empty(27, cin, cout, device=dev)
        print(f'== {X}x{Y}x{Z}  {cin}->{cout}  stride {st}: {flop / 1e9:.0f} GFLOP per direction', flush=True)

        def line(tag, fn):
            med, best = timed(fn, args.reps)
            print(f'  {tag:46s} {med:8.3f} ms  {flop / med / 1e9:7.1f} TF/s   (best {best:.3f} ms)', flush=True)
            return med
        # ---- forward
        nfm = int(hip.raw('es_spconv_split_workspace_floats')(n_out, 27, cin, cout))
        wsm = torch.zeros(max(nfm, 4), device=dev)
        if nfm:
            line('fwd   map kernel (split)', lambda: call('es_spconv_fwd_bf16_ws', P(xh), 1, cin, P(wt), P(nbr), n_out, n_in, 27, cin, cout, 0,
                                                          P(y), cout, 0, P(wsm), nfm, st_))
        else:
            line('fwd   map kernel', lambda: call('es_spconv_fwd_bf16', P(xh), 1, cin, P(wt), P(nbr), n_out, n_in, 27, cin, cout, 0, P(y), cout, 0, st_))
        variants = [(0, 1, 0), (256, 1, 0), (320, 1, 0), (0, 0, 0), (256, 1, 1), (320, 1, 1), (0, 1, 2), (0, 1, 3), (0, 1, 4), (0, 1, 6), (0, 1, 9)]
        for rows, order, split in variants:
            opt(20, rows); opt(21, order); opt(22, split)
            nf = int(hip.raw('es_dconv_workspace_floats')(g, 0, cin, cout))
            ws = torch.empty(max(nf, 4), device=dev)
            line(f'fwd   dense rows={rows} order={order} split={split}',
                 lambda: call('es_dconv_fwd_bf16', P(xh), cin, P(wt), g, 0, cin, cout, P(y), cout, 0, P(ws), nf, st_))
        opt(20, 0); opt(21, 0); opt(22, 0)
        # ---- data gradient
        nfm = int(hip.raw('es_spconv_split_workspace_floats')(n_in, 27, cout, cin))
        wsm = torch.zeros(max(nfm, 4), device=dev)
        if nfm:
            line('dgrad map kernel (split)', lambda: call('es_spconv_fwd_bf16_ws', P(dyh), 1, cout, P(wn), P(inv), n_in, n_out, 27, cout, cin, 0,
                                                          P(dx), cin, 0, P(wsm), nfm, st_))
        else:
            line('dgrad map kernel', lambda: call('es_spconv_fwd_bf16', P(dyh), 1, cout, P(wn), P(inv), n_in, n_out, 27, cout, cin, 0, P(dx), cin, 0, st_))
        if hip.raw('es_dconv_supported')(g, 1, cin, cout) == 1:
            nf = int(hip.raw('es_dconv_workspace_floats')(g, 1, cin, cout))
            ws = torch.empty(max(nf, 4), device=dev)
            line('dgrad dense (auto)', lambda: call('es_dconv_fwd_bf16', P(dyh), cout, P(wn), g, 1, cin, cout, P(dx), cin, 0, P(ws), nf, st_))
        # ---- weight gradient
        need = int(hip.raw('es_spconv_wgrad_workspace_floats')(1, P(xh), 1, cin, P(dyh), 1, cout, n_out, n_in, 27, cin, cout))
        wsw = torch.empty(max(need, 4), device=dev)
        line('wgrad map kernel', lambda: call('es_spconv_wgrad_bf16_src', P(xh), 1, cin, P(dyh), 1, cout, P(nbr), n_out, n_in, 27, cin, cout,
                                              P(dw), 0, P(wsw), need, st_))
        line('wgrad dense', lambda: call('es_dconv_wgrad_bf16', P(xh), cin, P(dyh), cout, g, 0, cin, cout, P(dw), 0, st_))
        del grid, nbr, inv, x, w, xh, wt, wn, dy, dyh, y, dx, dw
        torch.cuda.empty_cache()


if __name__ == '__main__':
    main()
