#!/usr/bin/env python
"""bench.py -- mv-3ddet train-step throughput on MI355X (BASELINE.json metric: scans/sec).

One "step" = one full train step of SparseFeatureFusionSingleStage3DDetector on a batch of synthetic
20 x (480x640) RGB-D scans already resident in HBM: depth->points (A1-A3), image normalisation (A18), 2-D and
3-D backbones, projection fusion, FCAF3D head, target assignment, losses, backward, gradient all-reduce
(N > 1), clip + AdamW.  Prints ONE JSON line (rank 0).

  python bench.py --gpus 1 --steps 5 --warmup 2
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
         bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=8)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=4, help='scans per GPU per step (reference config: 8xb4)')
    ap.add_argument('--views', type=int, default=20)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--precision', default='bf16', choices=['bf16', 'f32'],
                    help='conv fwd/dgrad matrix-core type: bf16 MFMA with f32 accumulate (BASELINE config) or exact-f32 MFMA')
    ap.add_argument('--cpu-views', type=int, default=20)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run'
    n_dev = torch.cuda.device_count()
    torch.cuda.set_device(local_rank % n_dev)
    dev = torch.device('cuda', local_rank % n_dev)
    if world > 1:
        # "nccl" == RCCL on ROCm.  ES_DIST_BACKEND=gloo lets the N>1 code path be exercised on a single-GPU box
        # (several ranks sharing one device), which RCCL refuses.
        backend = os.environ.get('ES_DIST_BACKEND', 'nccl')
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group(backend)

    from embodiedscan_amd import engine as E, hip, pipeline
    from embodiedscan_amd.config import build_detector, build_optim_wrapper, load_config
    from embodiedscan_amd.synth import make_scan

    E.PRECISION[0] = args.precision
    cfg = load_config(os.path.join(ROOT, 'configs', 'mv_3ddet.py'))
    det = build_detector(cfg, device=dev, seed=0).to(dev)          # same initial weights on every rank
    optim = build_optim_wrapper(cfg)

    # per-rank synthetic scans: seed = 1234 + rank*10007 + i ; rendered on the GPU, untimed
    scans = [make_scan(1234 + rank * 10007 + i, n_views=args.views, render_device=str(dev)) for i in range(args.batch)]
    dscans = [pipeline.upload_scan(s, dev) for s in scans]        # inputs resident in HBM before timing

    def step():
        batch = pipeline.make_batch(dscans)                       # A1-A3 on device
        return det.train_step(batch, optim)

    for _ in range(args.warmup):
        losses = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    # time EXACTLY `steps` steps; the convolution engine launches are bracketed by HIP events recorded on the stream
    # each kernel is launched on (the step runs on four streams: point branch, image branch, and their wgrad streams)
    prof = {'names': {'es_spconv_fwd', 'es_spconv_fwd_bf16', 'es_spconv_fwd_bf16_affine', 'es_spconv_wgrad',
                      'es_spconv_wgrad_bf16'}, 'records': [],
            'event': lambda: torch.cuda.Event(enable_timing=True)}
    t0 = time.perf_counter()
    for it in range(args.steps):
        # the engine launches of the LAST timed step are bracketed by HIP events (2 events per launch and one pair
        # counter per kernel map cost ~3 ms of host time per step, so they are not recorded on every step)
        hip.PROFILE = prof if it == args.steps - 1 else None
        losses = step()
    recs = prof['records']
    for i in range(len(recs)):              # resolve map pointer -> pair counter while the maps are still alive
        name, e0, e1, a = recs[i]
        recs[i] = (name, e0, e1, a, hip.PAIRS.get(a[4] if (name.startswith('es_spconv_wgrad') or name == 'es_spconv_fwd_bf16') else a[3]))
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    hip.PROFILE = None
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    in_sync = True
    rank_ms = None
    if world > 1:
        # per-rank time of the timed region (voxel counts differ per rank -> stragglers; SURVEY 8e asks for the spread)
        every = [torch.zeros(1, dtype=torch.float64, device=dev) for _ in range(world)]
        dist.all_gather(every, tmax.clone())
        per = sorted(float(t.item()) / args.steps * 1e3 for t in every)
        rank_ms = dict(min=round(per[0], 3), median=round(per[len(per) // 2], 3), max=round(per[-1], 3))
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        # data-parallel sanity (outside the timed region): every replica must hold the same parameters
        chk = det.arena.data[:det.arena.n_train].double().abs().sum().reshape(1)
        lo, hi = chk.clone(), chk.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        in_sync = bool(((hi - lo) <= 1e-9 * hi.abs()).item())
    dt = float(tmax.item())

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    # ---- roofline of the dominant kernel family (the convolution engine), from the live HIP-event timings.
    # The PMC passes (profiles/r1_conv_pmc*.txt) show the engine moving ~9 TB/s between L2 and the CUs at 16 % MFMA
    # utilisation: it is bandwidth bound, so the roofline is quoted against HBM bandwidth with the ALGORITHMIC bytes of
    # sum over launches of P*(Cin + Cout)*4 + weights, P = valid (output, tap) pairs of the kernel map.
    K_PEAK_HBM = 8000.0               # GB/s, MI355X_MICROARCH.md

    def engine_totals(records):
        tot_ms, tot_flop, tot_bytes, n_launch = 0.0, 0.0, 0.0, 0
        for name, e0, e1, a, pairs_dev in records:
            tot_ms += e0.elapsed_time(e1)
            n_launch += 1
            if name == 'es_spconv_fwd_bf16':
                nbr, n_out, n_in, K, cin, cout = a[4], a[5], a[6], a[7], a[8], a[9]
            elif not name.startswith('es_spconv_wgrad'):
                nbr, n_out, n_in, K, cin, cout = a[3], a[4], a[5], a[6], a[7], a[8]
            else:
                nbr, n_out, n_in, K, cin, cout = a[4], a[5], a[6], a[7], a[8], a[9]
            pairs = float(pairs_dev.item()) if pairs_dev is not None else (float(min(n_out, n_in)) if not nbr else float(n_out) * K)
            tot_flop += 2.0 * pairs * cin * cout
            wbytes = 2 if 'bf16' in name and not name.startswith('es_spconv_wgrad') else 4
            tot_bytes += pairs * (cin + cout) * 4.0 + float(K) * cin * cout * wbytes
        return tot_ms, tot_flop, tot_bytes, n_launch

    tot_ms, tot_flop, tot_bytes, n_launch = engine_totals(prof['records'])
    # One extra, UNTIMED step on the single-stream schedule: the same launches without other streams sharing the chip,
    # i.e. the stand-alone duration of each kernel (what a per-kernel roofline is usually quoted on).
    single = None
    if world == 1:
        from embodiedscan_amd import engine as E
        saved = (E.TWO_STREAMS[0], E.WGRAD_ASYNC[0])
        E.TWO_STREAMS[0] = E.WGRAD_ASYNC[0] = False
        prof1 = dict(prof, records=[])
        hip.PROFILE = prof1
        step()
        hip.PROFILE = None
        E.TWO_STREAMS[0], E.WGRAD_ASYNC[0] = saved
        r1 = prof1['records']
        r1 = [(n_, e0, e1, a, hip.PAIRS.get(a[4] if (n_.startswith('es_spconv_wgrad') or n_ == 'es_spconv_fwd_bf16') else a[3]))
              for n_, e0, e1, a in r1]
        torch.cuda.synchronize()
        ms1, fl1, by1, nl1 = engine_totals(r1)
        if ms1 > 0:
            single = dict(achieved=round(by1 / (ms1 * 1e-3) / 1e9, 1), frac=round(by1 / (ms1 * 1e-3) / 1e9 / K_PEAK_HBM, 4),
                          kernel_ms_per_step=round(ms1, 3), algorithmic_tflops=round(fl1 / (ms1 * 1e-3) / 1e12, 2),
                          note='same launches, one extra untimed step with ES_TWO_STREAMS=0 ES_WGRAD_ASYNC=0')
    achieved = tot_bytes / (tot_ms * 1e-3) / 1e9 if tot_ms > 0 else 0.0
    # HBM traffic of the same kernel family from the PMC passes of this command (FETCH_SIZE / WRITE_SIZE need separate
    # rocprofv3 runs, so the figure is read from the committed summary, bytes per launch like `achieved`'s numerator)
    traffic, traffic_note = None, 'traffic: null (no profiles/r1_final_pmc_traffic.json)'
    pmc_file = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles', 'r1_final_pmc_traffic.json')
    if args.precision == 'bf16' and os.path.exists(pmc_file):
        pmc = json.load(open(pmc_file))
        traffic = pmc['bytes_per_launch']
        traffic_note = (f"traffic = HBM bytes per launch from PMC (profiles/r1_final_pmc_traffic.json: "
                        f"{pmc['bytes_per_step'] / 1e9:.1f} GB per step over {pmc['launches'] // pmc['steps']} launches) vs "
                        f"{tot_bytes / max(n_launch, 1) / 1e6:.0f} MB algorithmic bytes per launch")
    roofline = dict(bound='hbm', achieved=round(achieved, 1), peak=K_PEAK_HBM, unit='GB/s',
                    frac=round(achieved / K_PEAK_HBM, 4), traffic=traffic,
                    kernel='convolution engine: k_spconv_bf16* (fwd/dgrad) + k_spconv_wgrad_bf16*' if args.precision == 'bf16'
                    else 'convolution engine: k_spconv / k_spconv_wgrad (exact-f32 MFMA)',
                    launches_per_step=n_launch, kernel_ms_per_step=round(tot_ms, 3), single_stream=single,
                    algorithmic_tflops=round(tot_flop / (tot_ms * 1e-3) / 1e12, 2) if tot_ms > 0 else 0.0,
                    note='algorithmic bytes = sum over launches of P*(Cin+Cout)*4 + K*Cin*Cout*sizeof(w), P = valid '
                         '(output,tap) pairs; algorithmic flops = 2*P*Cin*Cout; launch durations are HIP-event times on the '
                         'launch stream under the concurrent 4-stream schedule (kernels of different streams share the chip, '
                         'so the sum exceeds wall time); ' + traffic_note)

    out = dict(metric='scans/sec (train step) mv-3ddet, 20x(480x640) RGB-D views', value=round(world * args.batch * args.steps / dt, 4),
               unit='scans/s', n_gpus=world, steps=args.steps, warmup=args.warmup,
               ms_per_step=round(dt / args.steps * 1e3, 3), higher_is_better=True, scaling='weak', vs_baseline=None,
               dtype=args.precision, data='synthetic',
               config=dict(workload='mv-3ddet ResNet-50(w16) + MinkResNet34 + FCAF3DHeadRotMat, 20 views 480x640, '
                                    f'100k points/scan, {args.precision} matrix cores with f32 accumulate / f32 master weights, full train step incl. AdamW',
                           scans_per_gpu_per_step=args.batch, views=args.views, parallelism=f'dp{world}'),
               losses={k: round(float(v), 6) for k, v in losses.items()}, roofline=roofline)
    if world > 1:
        out['replicas_in_sync'] = in_sync
        out['rank_ms_per_step'] = rank_ms
    if world == 1 and not args.no_cpu_baseline:
        out['cpu_baseline'] = cpu_baseline(scans[0], det, args)
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def cpu_baseline(scan, det, args):
    """The CPU oracle (a restatement, kind='port') timed on this box's host cores on ONE scan of the same workload:
    forward + backward of the detector loss (no optimiser).  Bounded sample, reported next to the GPU number."""
    import torch
    from oracle import model as OM, pipeline as OP
    sd = {k: v.cpu() for k, v in det.state_dict().items()}
    names = set(det.arena.grad_dict().keys())
    sd = {k: v.requires_grad_(k in names) for k, v in sd.items()}
    t0 = time.perf_counter()
    pts = [OP.scan_to_points(scan)]
    imgs = OM.preprocess_img(torch.from_numpy(scan['img']), [123.675, 116.28, 103.53], [58.395, 57.12, 57.375])[None]
    losses = OM.detector_loss(sd, pts, imgs, [scan['meta']], [torch.from_numpy(scan['gt_boxes'])],
                              [torch.from_numpy(scan['gt_labels'])])
    sum(losses.values()).backward()
    dt = time.perf_counter() - t0
    return dict(value=round(1.0 / dt, 5), unit='scans/s', cores=torch.get_num_threads(), kind='port',
                sample=f'1 scan x {scan["depth"].shape[0]} views 480x640, 100k points, one forward+backward of the '
                       f'PyTorch-f32 CPU oracle (no optimiser step), {dt:.1f} s; os.cpu_count()={os.cpu_count()}')


if __name__ == '__main__':
    main()
