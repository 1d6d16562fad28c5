#!/usr/bin/env python
"""bench.py -- mv-3ddet train-step throughput on MI355X (BASELINE.json metric: scans/sec).

One "step" = one full train step of SparseFeatureFusionSingleStage3DDetector on a batch of synthetic
20 x (480x640) RGB-D scans: host->device copy of the raw batch (pinned buffers, copy stream, double-buffered slots:
the copy of step i+1 runs under the kernels of step i, the step waits for its own batch) -> depth->points (A1-A3) ->
image normalisation (A18) -> 2-D and 3-D backbones -> projection fusion -> FCAF3D head -> target assignment -> losses ->
backward -> gradient all-reduce (N > 1) -> clip + AdamW.  The timed region rotates through `--rotate` distinct batches.
Prints ONE JSON line (rank 0).

  python bench.py --gpus 1 --steps 8 --warmup 3
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
# the host driver only supports dmabuf IPC: RCCL's peer buffers fail with hipIpcGetMemHandle errors without it
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

K_PEAK_HBM = 8000.0               # GB/s      (MI355X_MICROARCH.md)
K_PEAK_MFMA = {'bf16': 2500.0, 'f32': 157.3}    # dense TFLOP/s of the matrix-core type the engine computes in
ENGINE = {'es_spconv_fwd', 'es_spconv_fwd_bf16', 'es_spconv_fwd_bf16_ws', 'es_spconv_fwd_bf16_affine', 'es_spconv_fwd_bf16_io',
          'es_spconv_wgrad', 'es_spconv_wgrad_bf16', 'es_spconv_wgrad_bf16_src', 'es_dconv_fwd_bf16',
          'es_dconv_wgrad_bf16', 'es_dconv_wgrad_ws_bf16', 'es_spconv_halo_bf16', 'es_img_wgrad9_bf16', 'es_img_conv3_bf16', 'es_rows_wgrad1_bf16'}
IMGC = ('es_img_conv3_bf16',)        # round 6: (X, ldx, W, n_img, H, W, C, stride, mode, scale, shift, gate, ldg, act, Y, y_half, ldy, stream)
ROWW = ('es_rows_wgrad1_bf16',)      # round 6: (Xh, ldx, dY, ldy, n, Cin, Cout, dW, acc, ws, ws_floats, stream)
IMGW = ('es_img_wgrad9_bf16',)       # round 6: (Xh, ldx, dY, ldy, n_img, H, W, C, stride, dW, acc, ws, ws_floats, stream)
HALO = ('es_spconv_halo_bf16',)      # round 6: (Xh, ldx, W, loc, hrows, hcnt, n_out, n_in, K, Cin, Cout, bias, Y, ldy, acc, stream)
DENSE = ('es_dconv_fwd_bf16', 'es_dconv_wgrad_bf16', 'es_dconv_wgrad_ws_bf16')   # round 5: the dense-volume engine (csrc/dconv.hip)
DENSE_WGRAD = ('es_dconv_wgrad_bf16', 'es_dconv_wgrad_ws_bf16')
FWD_X = ('es_spconv_fwd_bf16', 'es_spconv_fwd_bf16_ws', 'es_spconv_fwd_bf16_io')     # (X, x_half, ldx, W, nbr, n_out, n_in, K, Cin, Cout, ...)
SCATTER = {'es_voxel_keys', 'es_unique_first', 'es_morton_sort', 'es_stride_keys', 'es_kernel_map', 'es_inverse_map',
           'es_union_plan', 'es_point_sample_fwd', 'es_point_sample_bwd', 'es_depth_to_points'}


_COPY_STREAM = {}


def copy_stream(dev):
    """ONE copy stream per device for the whole process: every configuration of the default run feeds through it.  (A fresh
    torch.cuda.Stream() per leg lands on another of the 4 hardware queues each time; whichever leg ran second then shared a queue
    between two of its streams and measured ~10 % slower -- from_files behind grounding 131 vs 157-160 scans/s, grounding behind
    from_files 62 vs 56.6 ms: profiles/r5k_*, r5_bench_default.json.)"""
    import torch
    key = str(dev)
    if key not in _COPY_STREAM:
        _COPY_STREAM[key] = torch.cuda.Stream()
    return _COPY_STREAM[key]


class Feeder:
    """Host->device feed of whole batches: every batch is ONE pinned slab (pipeline.pin_batch), the device side two byte
    slabs; the copy of step i+1 is queued on a copy stream under the kernels of step i (one hipMemcpyAsync per batch),
    the step waits for its own batch and the slot is recycled when the step that read it has been queued completely."""

    def __init__(self, batches, dev, resident=False):
        import torch
        from embodiedscan_amd import pipeline
        self.torch, self.pipeline = torch, pipeline
        self.batches, self.dev, self.resident = batches, dev, resident
        self.h2d_bytes = batches[0].nbytes
        cap = max(b.nbytes for b in batches)
        self.slots = [pipeline.alloc_batch_slot(cap, dev) for _ in range(2)]
        self.copy_stream = copy_stream(dev)
        self.ready = [torch.cuda.Event(), torch.cuda.Event()]     # slot s holds its batch
        self.freed = [torch.cuda.Event(), torch.cuda.Event()]     # the step that read slot s has been queued completely
        self.i = 0
        self.dscans = [None, None]

    def _prefetch(self, i):
        s = i % 2
        with self.torch.cuda.stream(self.copy_stream):
            if i >= 2:
                self.copy_stream.wait_event(self.freed[s])        # step i-2 (last reader of this slot) is done
            self.dscans[s] = self.pipeline.upload_batch(self.slots[s], self.batches[i % len(self.batches)])
            self.ready[s].record(self.copy_stream)

    def next(self):
        """dscans of this step's batch (the current stream is made to wait for its copy; the next copy is queued)"""
        i, s = self.i, self.i % 2
        if self.resident:
            if self.dscans[0] is None:
                self.dscans[0] = self.pipeline.upload_batch(self.slots[0], self.batches[0])
            return self.dscans[0]
        if i == 0:
            self._prefetch(0)
        self.torch.cuda.current_stream().wait_event(self.ready[s])
        self._prefetch(i + 1)
        return self.dscans[s]

    def done(self):
        if not self.resident:
            self.freed[self.i % 2].record(self.torch.cuda.current_stream())
        self.i += 1


def stage_times(step_fn, runs=3):
    """per-stage HIP-event times (engine.mark boundaries) of `runs` warmed steps on the CURRENT schedule -> median per stage"""
    import statistics
    import torch
    from embodiedscan_amd import engine as E
    step_fn()                                                   # warm this schedule (allocator blocks, eager image backbone)
    torch.cuda.synchronize()
    per = []
    for _ in range(runs):
        E.MARKS = []
        step_fn()
        marks, E.MARKS = E.MARKS, None
        torch.cuda.synchronize()
        st = {}
        for (n0, ev0), (n1, ev1) in zip(marks[:-1], marks[1:]):
            st[n1] = st.get(n1, 0.0) + ev0.elapsed_time(ev1)
        per.append(st)
    return {k: round(statistics.median(p.get(k, 0.0) for p in per), 3) for k in per[0]}


def release_frozen():
    """between configurations: the detector that was just dropped froze the collector's generations after its warm-up
    (BaseDetector._settle_gc); thaw them and collect once so that whatever it held in reference cycles goes back to the allocator"""
    import gc
    import torch
    gc.unfreeze()
    gc.collect()
    torch.cuda.empty_cache()


def self_launch(n):
    """re-exec this command under torch.distributed.run with n local ranks (rendezvous on 127.0.0.1, a free port)"""
    import socket
    with socket.socket() as so:
        so.bind(('127.0.0.1', 0))
        port = so.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    os.execvpe(cmd[0], cmd, env)


def dry_launch(rank, world):
    """the launch / rendezvous / collective plumbing of the N > 1 path with no GPU: what tests/test_host_logic.py runs here"""
    import torch
    import torch.distributed as dist
    backend = os.environ.get('ES_DIST_BACKEND', 'gloo')
    if world > 1:
        dist.init_process_group(backend)
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t)
        ranks = dist.get_world_size()
    else:
        ranks = 1
    ok = abs(float(t.item()) - world * (world + 1) / 2) < 1e-12
    if rank == 0:
        print(json.dumps(dict(dry_launch=True, n_gpus=world, ranks=ranks, dist_backend=backend if world > 1 else None,
                              allreduce_ok=ok, master=f"{os.environ.get('MASTER_ADDR')}:{os.environ.get('MASTER_PORT')}")))
    if world > 1:
        dist.destroy_process_group()
    if not ok:
        raise SystemExit('dry launch: all-reduce of the rank tokens is wrong')


LINE_LIMIT = 4096                     # the driver keeps the tail of stdout: the LAST line must stay well under that (a 21.8 KB line was lost)


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d and d[k] is not None}


def _parity_short(p):
    if not p:
        return None
    errs = [v for v in (p.get('rel_err') or {}).values() if isinstance(v, (int, float))]
    return dict(ok=bool(p.get('ok')), max_rel_err=max(errs) if errs else None, tol=p.get('tol'))


def _roofline_short(r, keys=('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'kernel', 'kernel_ms_per_step', 'launches_per_step',
                             'durations', 'frac_standalone', 'dominant')):
    if not r:
        return None
    out = _pick(r, keys)
    out.setdefault('traffic', None)
    if isinstance(out.get('kernel'), str) and len(out['kernel']) > 120:
        out['kernel'] = out['kernel'][:117] + '...'
    return out


def compact_line(out, detail=None):
    """The ONE line the driver parses: the contract's keys + roofline + cpu_baseline + parity, every other config reduced to
    {value, ms_per_step, roofline_frac, parity_ok}; class tables, stage times, per-step lists, host blocks, loss dumps and notes
    live in the detail file (`bench_detail.json`).  Guaranteed < LINE_LIMIT bytes: optional keys are dropped from the back if a
    future field ever grows (tests/test_bench_line.py)."""
    line = _pick(out, ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling'))
    line['vs_baseline'] = out.get('vs_baseline')
    line.update(_pick(out, ('dtype', 'data')))
    cfg = out.get('config') or {}
    line['config'] = _pick(cfg, ('workload', 'scans_per_gpu_per_step', 'views', 'parallelism', 'h2d'))
    if isinstance(line['config'].get('workload'), str) and len(line['config']['workload']) > 260:
        line['config']['workload'] = line['config']['workload'][:257] + '...'
    line['roofline'] = _roofline_short(out.get('roofline'))
    cb = out.get('cpu_baseline')
    if cb:
        cb = _pick(cb, ('value', 'unit', 'cores', 'kind', 'sample'))
        if isinstance(cb.get('sample'), str) and len(cb['sample']) > 200:
            cb['sample'] = cb['sample'][:197] + '...'
    line['cpu_baseline'] = cb
    line['parity'] = _parity_short(out.get('parity'))
    line['ranks'] = out.get('ranks', out.get('n_gpus', 1))
    line['dist_backend'] = out.get('dist_backend')
    optional = []
    for k in ('replicas_in_sync', 'rank_ms_per_step'):
        if k in out:
            line[k] = out[k]
            optional.append(k)
    if isinstance(out.get('allreduce_exposed_ms'), dict):
        line['allreduce_exposed_ms'] = out['allreduce_exposed_ms'].get('per_part')
        optional.append('allreduce_exposed_ms')
    sc = out.get('scatter_path')
    if sc:
        line['scatter_path'] = _pick(sc, ('achieved_GBps', 'peak_GBps', 'frac', 'ms_per_step'))
        optional.append('scatter_path')
    oc = out.get('other_configs')
    if oc:
        line['other_configs'] = {}
        for kind, r in oc.items():
            if r.get('error'):
                line['other_configs'][kind] = dict(error=str(r['error'])[:160])
                continue
            e = dict(value=r.get('value'), ms_per_step=r.get('ms_per_step'))
            if r.get('roofline'):
                e['roofline_frac'] = r['roofline'].get('frac')
            if r.get('parity'):
                e['parity_ok'] = bool(r['parity'].get('ok'))
            if 'vs_synthetic' in r:
                e['vs_synthetic'] = r['vs_synthetic']
            line['other_configs'][kind] = e
    if detail:
        line['detail'] = detail
    # belt and braces: never emit a line the driver cannot keep
    for k in ['detail'] + optional[::-1] + ['other_configs']:
        if len(json.dumps(line)) < LINE_LIMIT:
            break
        line.pop(k, None)
    return line


def write_detail(out, name='bench_detail.json'):
    """every figure of the run (class tables, stage times, per-step lists, host block, loss dumps, notes) goes to a file next to the
    script and, when it exists, under gpurun_out/ (the directory that travels back from a GPU box); returns the repo-relative path"""
    paths = [os.path.join(ROOT, name)]
    if os.path.isdir(os.path.join(ROOT, 'gpurun_out')):
        paths.append(os.path.join(ROOT, 'gpurun_out', name))
    wrote = None
    for p in paths:
        try:
            with open(p, 'w') as f:
                json.dump(out, f)
            wrote = wrote or os.path.relpath(p, ROOT)
        except OSError:
            pass
    return wrote


def emit(out, name='bench_detail.json'):
    """detail to the file(s), the short line as the LAST line of stdout"""
    detail = write_detail(out, name)
    line = compact_line(out, detail)
    sys.stdout.flush()
    print(json.dumps(line), flush=True)
    return line



def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=8)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=4, help='scans per GPU per step (reference config: 8xb4)')
    ap.add_argument('--views', type=int, default=20)
    ap.add_argument('--rotate', type=int, default=3, help='distinct synthetic batches the timed steps cycle through')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--precision', default='bf16', choices=['bf16', 'f32'],
                    help='conv fwd/dgrad matrix-core type: bf16 MFMA with f32 accumulate (BASELINE config) or exact-f32 MFMA')
    ap.add_argument('--resident', action='store_true', help='skip the per-step host->device copy (inputs resident in HBM)')
    ap.add_argument('--no-other-configs', action='store_true',
                    help='skip BASELINE configs 4 (mv-grounding) and 5 (occupancy), which the default 1-GPU run appends as `other_configs`')
    ap.add_argument('--only', default=None, choices=['grounding', 'occupancy', 'from_files'],
                    help='run ONLY that configuration and print its object as the JSON line (profiling passes)')
    ap.add_argument('--grounding-batch', type=int, default=12, help='scans per GPU per step of config 4 (reference: 8xb12)')
    ap.add_argument('--other-steps', type=int, default=10, help='timed steps of the other configs (capped by --steps)')
    ap.add_argument('--dry-launch', action='store_true',
                    help='N > 1 plumbing check without a GPU: every rank joins the process group (ES_DIST_BACKEND, default gloo here), '
                         'all-reduces a token, rank 0 prints one JSON line')
    args = ap.parse_args()

    # `python bench.py --gpus N` with N > 1 and no launcher around it (WORLD_SIZE unset): become the launcher -- one rank per GPU
    # under torch.distributed.run on 127.0.0.1 (the container hostname may not resolve), same arguments, same stdout.  The
    # documented form `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` arrives with WORLD_SIZE set.
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        self_launch(args.gpus)                                      # does not return
    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: run `python bench.py --gpus {args.gpus}` (self-launching) or '
                         f'`python -m torch.distributed.run --nnodes=1 --nproc-per-node {args.gpus} --master-addr 127.0.0.1 bench.py --gpus {args.gpus}`')
    if args.dry_launch:
        return dry_launch(rank, world)

    import torch
    import torch.distributed as dist
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

    if args.only:
        assert world == 1
        E.PRECISION[0] = args.precision
        res = run_from_files(args, dev) if args.only == 'from_files' else run_other_config(args.only, args, dev)
        emit(res, f'bench_detail_{args.only}.json')
        if res.get('parity') and not res['parity']['ok']:
            raise SystemExit(f'parity check of config {args.only} FAILED: ' + json.dumps(res['parity']))
        return

    # (Measured and rejected: running the step's dependent chain on a high-priority stream -- 42-45 ms/step against 30.7 on
    # the same box; the weight-gradient / side streams starve behind it and the joins at the end of backward wait longer.)
    E.PRECISION[0] = args.precision
    cfg = load_config(os.path.join(ROOT, 'configs', 'mv_3ddet.py'))
    det = build_detector(cfg, device=dev, seed=0).to(dev)          # same initial weights on every rank
    optim = build_optim_wrapper(cfg)

    # per-rank synthetic scans: seed = 1234 + rank*10007 + i ; rendered on the GPU, untimed.  `rotate`
    # distinct batches live in PINNED host memory; every timed step copies its batch host->device.
    n_rot = max(1, args.rotate)
    scans = [make_scan(1234 + rank * 10007 + i, n_views=args.views, render_device=str(dev))
             for i in range(args.batch * n_rot)]
    feeder = Feeder([pipeline.pin_batch(scans[r * args.batch:(r + 1) * args.batch]) for r in range(n_rot)], dev,
                    resident=args.resident)
    h2d_bytes = feeder.h2d_bytes

    # ---- parity at the benchmarked configuration, part 1 (untimed): losses of scans[0] alone from the PRE-training
    # weights on the HIP path; cpu_baseline() below runs the oracle on the same scan / weights and asserts agreement
    parity_hip = None
    sd0 = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        sd0 = {k: v.cpu() for k, v in det.state_dict().items()}
        d0 = pipeline.upload_scan(scans[0], dev)
        b0 = pipeline.make_batch([d0])
        E.TAPE.clear()
        data = det.data_preprocessor(b0, True)
        det._bind()
        l0 = det.forward(data['inputs'], data['data_samples'], mode='loss')
        E.TAPE.clear()
        E.join_wgrad_streams()
        parity_hip = {k: float(v) for k, v in l0.items()}
        del d0, b0, data, l0

    # Software pipeline over batches (round 4): right after train_step(i) has been queued, the WEIGHT-INDEPENDENT prefix of step
    # i+1 -- wait for its H2D copy, A1-A3 depth -> points, image normalisation, A4 voxelisation + Morton order, the strided
    # chain with its row-count read-backs, the 3-D backbone's / head's kernel maps and unions (A6) -- is issued on a
    # high-priority side stream (det.prefetch), i.e. under step i's backward pass; what the reference does with DataLoader
    # workers on host cores.  Every timed step prefetches for its successor (the first timed step's prefix was issued by the
    # last warm-up step, the last timed step issues the prefix of a step that is not timed): K full steps of work in the
    # timed region.  MEASURED AND REJECTED as the default (round 4, profiles/r4g_host_profile_*.txt, r4h/r4i sweeps): the host needs
    # only ~9 ms to queue a step, so the device was never waiting for it; with the prefetch chain (~220 small launches + five
    # host round trips) running beside the step, the 3-D backbone's forward -- itself a chain of small dependent launches --
    # slows from 3.2 to 7-10 ms and the backward from 14 to 18-20 ms: 39.5 ms / step against 26.5 serial, whatever the stream
    # priority, the number of hardware queues, or a gate that holds the prefetch back until the 3-D forward has finished.
    # ES_NEXT_PREFETCH=1 enables it (bit-identical results: tests/test_gpu_prefetch.py).
    nxt = [None]
    use_prefetch = os.environ.get('ES_NEXT_PREFETCH', '0') == '1' and hasattr(det, 'prefetch')

    def make():
        return pipeline.make_batch(feeder.next())             # this batch has landed in HBM (next copy queued); A1-A3 on device

    def step(prefetch_next=True):
        E.mark('_begin')
        batch, nxt[0] = (nxt[0] if nxt[0] is not None else make()), None
        out = det.train_step(batch, optim)
        feeder.done()
        if use_prefetch and prefetch_next:
            nxt[0] = det.prefetch(make)
        return out

    # four priming steps that are NOT part of --warmup: the preprocessor alternates between two output buffers, and for each of
    # them the first step builds the lazily made maps / allocator blocks and the second captures the image backbone's launch
    # sequence into a hipGraph (engine.graphed: one graph per input address) -- so that even `--warmup 0` times steady steps
    for _ in range(4):
        step()
    for _ in range(args.warmup):
        losses = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    # time EXACTLY `steps` steps; the convolution engine launches of ONE EXTRA step after them are bracketed by HIP events
    # recorded on the stream each kernel is launched on (the step runs on four compute streams + the copy stream)
    prof = {'names': ENGINE, 'records': [], 'event': lambda: torch.cuda.Event(enable_timing=True)}
    step_ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    step_ev[0].record()
    cfs0 = cfs_stat()
    t0 = time.perf_counter()
    for it in range(args.steps):
        losses = step()                                         # no instrumentation of any kind inside the timed region
        step_ev[it + 1].record()                                # end of the step's main-stream work (no sync)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    host = host_report(cfs0)
    # one extra UNTIMED step on the same (four-stream) schedule with every engine launch bracketed by HIP events (2 events per
    # launch and a pair counter per kernel map cost ~3 ms of host time, and instrumented launches cannot be replayed from the
    # image backbone's hipGraph -- so this step is kept out of the timed region since round 3)
    hip.PROFILE = prof
    red = getattr(det.arena, 'reducer', None)
    if red is not None:
        red.profile = []                                        # N > 1: how long the optimiser waits for each gradient bucket
    step(prefetch_next=False)                                   # (consumes the batch the last timed step prepared)
    recs = resolve_pairs(hip, prof['records'])
    torch.cuda.synchronize()
    hip.PROFILE = None
    if red is not None:
        prof_red, red.profile = red.profile, None
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
        # the clip norm was assembled from per-bucket sums of squares taken behind the all-reduces (optim.py): compare it with
        # the norm of the mean gradient computed in one piece (the arena still holds the SUM over ranks of the last step)
        ref_norm = float(det.arena.grad[:det.arena.n_train].double().pow(2).sum().sqrt().item()) * getattr(optim, 'last_gscale', 1.0)
        got_norm = float(optim.norm.item())
        in_sync = in_sync and abs(got_norm - ref_norm) <= 1e-4 * max(ref_norm, 1e-12)
    dt = float(tmax.item())

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    mfma_peak = K_PEAK_MFMA[args.precision]
    eng = engine_totals(recs, mfma_peak)

    # ---- one extra UNTIMED step on the single-stream schedule: stand-alone duration of every engine launch, the
    # scatter-path kernels (north_star: "achieved HBM GB/s for the scatter path") and per-stage times
    single = scatter = stages = classes = None
    if world == 1:
        saved = (E.TWO_STREAMS[0], E.WGRAD_ASYNC[0])
        E.TWO_STREAMS[0] = E.WGRAD_ASYNC[0] = False
        stages = stage_times(lambda: step(prefetch_next=False))     # median of 3 warmed serial steps (coordinate phase inside)
        prof1 = dict(prof, names=ENGINE | SCATTER, records=[])
        hip.PROFILE = prof1
        step(prefetch_next=False)                                   # + one with every engine / scatter launch bracketed
        hip.PROFILE = None
        E.TWO_STREAMS[0], E.WGRAD_ASYNC[0] = saved
        r1 = resolve_pairs(hip, prof1['records'])
        torch.cuda.synchronize()
        e1 = engine_totals([r for r in r1 if r[0] in ENGINE], mfma_peak)
        single = dict(kernel_ms_per_step=e1['ms'], launches=e1['launches'], algorithmic_tflops=e1['tflops'],
                      frac_of_binding_roof=e1['frac_binding'], pair_bytes_GBps=e1['pair_GBps'],
                      note='same launches, one extra untimed step with ES_TWO_STREAMS=0 ES_WGRAD_ASYNC=0 (stand-alone '
                           'durations); rocprofv3 summary of this schedule: profiles/r3_single_stream_kernel_stats.txt')
        scatter = scatter_totals([r for r in r1 if r[0] in SCATTER])
        classes = launch_classes(r1, mfma_peak)
        if os.environ.get('ES_BENCH_DUMP'):                     # dev: every engine launch of the single-stream step, in order
            with open(os.environ['ES_BENCH_DUMP'], 'w') as f:
                for name, ev0, ev1, a, _ in r1:
                    if name in ENGINE:
                        nbr, n_out, n_in, K, cin, cout = engine_args(name, a)
                        f.write(json.dumps(dict(fn=name, K=K, cin=cin, cout=cout, n_out=n_out, n_in=n_in, map=bool(nbr),
                                                us=round(ev0.elapsed_time(ev1) * 1e3, 1),
                                                args=[x for x in a if isinstance(x, int) and abs(x) < (1 << 31)])) + '\n')
        stages['_note'] = 'single-stream schedule (ES_TWO_STREAMS=0 ES_WGRAD_ASYNC=0): HIP-event time between stage boundaries, MEDIAN of 3 ' \
                          'untimed steps after one warm-up step on that schedule; includes host-induced gaps; the default four-stream ' \
                          'schedule overlaps A7 with A4-A6 and the two backward branches'

    # static PMC figures of the same command (separate rocprofv3 --pmc passes, see profiles/): bytes per engine launch
    traffic, traffic_extra, traffic_note = None, {}, 'traffic: null (no PMC summary for this precision under profiles/)'
    for name in ('r6_pmc_traffic.json', 'r5_pmc_traffic.json', 'r4_pmc_traffic.json', 'r3_pmc_traffic.json', 'r2_pmc_traffic.json', 'r1_final_pmc_traffic.json'):
        pmc_file = os.path.join(ROOT, 'profiles', name)
        if args.precision == 'bf16' and os.path.exists(pmc_file):
            pmc = json.load(open(pmc_file))
            traffic = pmc['bytes_per_launch']
            # `traffic` is per launch of the PMC family (the engine kernels AND their reduction helpers); multiply by ITS launch
            # count -- not by roofline.launches_per_step, which counts engine entry points -- to get the bytes per step
            traffic_extra = dict(traffic_launches_per_step=pmc['launches'] // pmc['steps'], traffic_bytes_per_step=pmc['bytes_per_step'],
                                 traffic_source=f'profiles/{name}')
            traffic_note = (f"traffic is STATIC: HBM bytes per launch of the convolution-engine kernel family from the committed PMC passes "
                            f"of this command (profiles/{name}: {pmc['bytes_per_step'] / 1e9:.1f} GB per step over "
                            f"{pmc['launches'] // pmc['steps']} kernel launches = traffic x traffic_launches_per_step), not re-measured by this run")
            break
    # achieved / frac / kernel_ms_per_step come from the STAND-ALONE durations (the extra single-stream step, = what a rocprofv3 kernel
    # trace of that schedule sums: profiles/r5_single_stream_kernel_stats.txt) when this is a one-GPU run: HIP-event brackets under the
    # four-stream schedule overlap in time and over-count (round-4 finding: 29.3 ms of "kernel time" in a 25.6 ms step); the concurrent
    # figures stay in `concurrent_schedule`
    # Round 6: ONE definition, kept from here on -- the family's launches on the DEFAULT (four-stream) schedule, HIP-event
    # duration of each launch on its own stream, summed (an upper bound of what a rocprofv3 kernel trace of the same schedule sums:
    # events also see the time a launch waits for a free CU); the stand-alone figure of the single-stream extra step stays beside it
    # as `frac_standalone`, the dominant launch class (stand-alone durations) as `dominant`.
    base = eng
    if base['t_mfma'] >= base['t_hbm']:
        roof = dict(bound='mfma', achieved=base['tflops'], peak=mfma_peak, unit='TFLOP/s',
                    frac=round(base['tflops'] / mfma_peak, 4))
    else:
        roof = dict(bound='hbm', achieved=base['comp_GBps'], peak=K_PEAK_HBM, unit='GB/s',
                    frac=round(base['comp_GBps'] / K_PEAK_HBM, 4))
    roofline = dict(roof, traffic=traffic, **traffic_extra,
                    kernel='convolution engine: k_spconv_* (gather / halo / narrow) / k_rowgemm2 / k_lin_small / k_expand / k_img_conv3 (fwd, dgrad) + k_spconv_wgrad_bf16* / k_img_wgrad9 / k_rows_wgrad1' if args.precision == 'bf16'
                    else 'convolution engine: k_spconv / k_spconv_wgrad (exact-f32 MFMA)',
                    launches_per_step=base['launches'], kernel_ms_per_step=base['ms'],
                    frac_of_binding_roof=base['frac_binding'],
                    durations='HIP events per launch, default four-stream schedule, one extra step after the timed ones',
                    frac_standalone=(None if single is None else round(
                        (e1['tflops'] / mfma_peak) if roof['bound'] == 'mfma' else (e1['comp_GBps'] / K_PEAK_HBM), 4)),
                    dominant=(None if not classes else dict(cls=classes[0]['cls'], launches=classes[0]['launches'], ms=classes[0]['ms'],
                                                           tflops=classes[0]['tflops'], frac_mfma=round(classes[0]['tflops'] / mfma_peak, 4),
                                                           frac_of_binding_roof=classes[0]['frac_of_binding_roof'])),
                    concurrent_schedule=dict(kernel_ms_per_step=eng['ms'], achieved_GBps=eng['comp_GBps'], tflops=eng['tflops'],
                                             frac_of_binding_roof=eng['frac_binding']),
                    mfma=dict(achieved_tflops=base['tflops'], peak_tflops=mfma_peak, frac=round(base['tflops'] / mfma_peak, 4)),
                    hbm_compulsory=dict(achieved_GBps=base['comp_GBps'], frac=round(base['comp_GBps'] / K_PEAK_HBM, 4),
                                        bytes_per_step=base['comp_bytes']),
                    hbm_pair_bytes=dict(achieved_GBps=base['pair_GBps'], frac=round(base['pair_GBps'] / K_PEAK_HBM, 4),
                                        bytes_per_step=base['pair_bytes']),
                    single_stream=single, classes=classes,
                    note='per launch: algorithmic flops = 2*P*Cin*Cout (P = valid (output,tap) pairs), compulsory bytes = '
                         'every input row, output row and weight once, pair bytes = SURVEY 8(d) P*(Cin+Cout)*4 + weights; '
                         'binding roof per launch = max(flops/MFMA peak, compulsory bytes/HBM peak); frac_of_binding_roof = '
                         'sum of binding-roof times / sum of HIP-event launch durations; bound/achieved/peak/frac = the '
                         'roof that binds the family in total; durations (see `durations`) are HIP-event times on the launch stream: stand-alone = '
                         'one extra single-stream step (round 5: the headline figures); under the '
                         'concurrent 4-stream schedule (kernels of different streams share the chip, so the sum exceeds wall '
                         'time), taken on ONE EXTRA untimed step right after the timed ones -- per-launch events cost ~3 ms of '
                         'host time and keep the image backbone off its hipGraph, so the timed steps carry no instrumentation; classes = the top engine launch classes of the single-stream step (stand-alone durations); '
                         + traffic_note)

    out = dict(metric='scans/sec (train step) mv-3ddet, 20x(480x640) RGB-D views', value=round(world * args.batch * args.steps / dt, 4),
               unit='scans/s', n_gpus=world, steps=args.steps, warmup=args.warmup,
               ms_per_step=round(dt / args.steps * 1e3, 3), higher_is_better=True, scaling='weak', vs_baseline=None,
               dtype=args.precision, data='synthetic',
               config=dict(workload='mv-3ddet ResNet-50(w16) + MinkResNet34 + FCAF3DHeadRotMat, 20 views 480x640, '
                                    f'100k points/scan, {args.precision} matrix cores with f32 accumulate / f32 master weights, '
                                    'full train step incl. H2D of the batch and AdamW',
                           scans_per_gpu_per_step=args.batch, views=args.views, parallelism=f'dp{world}',
                           distinct_batches=n_rot,
                           h2d='resident (no per-step copy)' if args.resident else
                           f'{h2d_bytes / 1e6:.0f} MB per step as ONE copy from a pinned slab on a copy stream, double-buffered'),
               losses={k: round(float(v), 6) for k, v in losses.items()}, roofline=roofline)
    if scatter is not None:
        out['scatter_path'] = scatter
        out['stage_ms'] = stages
    out['ranks'] = dist.get_world_size() if world > 1 else 1
    out['dist_backend'] = None                                   # one process, no process group
    if world > 1:
        out['dist_backend'] = dist.get_backend() + (' (RCCL)' if dist.get_backend() == 'nccl' else '')
        out['replicas_in_sync'] = in_sync
        out['rank_ms_per_step'] = rank_ms
        if red is not None and prof_red:
            exposed, mb = {}, {}
            for part, floats, e0, e1 in prof_red:
                exposed[part] = round(exposed.get(part, 0.0) + e0.elapsed_time(e1), 3)
                mb[part] = round(mb.get(part, 0.0) + floats * 4 / 2 ** 20, 1)
            out['allreduce_exposed_ms'] = dict(per_part={str(k): v for k, v in sorted(exposed.items())},
                                               MiB_per_part={str(k): v for k, v in sorted(mb.items())},
                                               note='rank 0, the extra instrumented step: time the compute stream stalls in the optimiser waiting for each '
                                                    'gradient part (0 = 2-D backbone, 1 = 3-D backbone, 2 = head; launched from tape markers '
                                                    'in backward-completion order 2, 1, 0; the clip norm is taken per bucket behind its all-reduce)')
    out['hipgraph'] = dict(E.GRAPH_STATS, what='image-backbone forward sequences (engine.graphed): captured / replayed / run eagerly')
    out['optimizer_pass'] = OPT_PASS.get(optim.last_path, optim.last_path)
    out['host'] = host
    # GPU-side duration of each timed step (events on the main stream)
    out['step_ms'] = [round(step_ev[i].elapsed_time(step_ev[i + 1]), 2) for i in range(args.steps)]
    if world == 1 and not args.no_cpu_baseline:
        out['cpu_baseline'], out['parity'] = cpu_baseline(scans[0], sd0, det, parity_hip, args)
    # ---- BASELINE configs 4 and 5 on the same GPU, same rules (H2D in the step, rotating batches, parity vs the oracle)
    bad = []
    if world == 1 and not args.no_other_configs:
        del det, optim, feeder, scans, losses
        E.TAPE.clear()
        release_frozen()
        out['other_configs'] = {}
        # (from_files first: behind the grounding leg it measured 131-132 scans/s twice, behind nothing or behind occupancy 157-160 -- profiles/r5k_*)
        for kind in [k for k in ('from_files', 'grounding', 'occupancy') if k in os.environ.get('ES_OTHER', 'grounding,occupancy,from_files').split(',')]:
            try:
                r = run_from_files(args, dev, out['value']) if kind == 'from_files' else run_other_config(kind, args, dev)
            except Exception as e:                              # the primary line must survive a failure here
                import traceback
                r = dict(error=f'{type(e).__name__}: {e}', traceback=traceback.format_exc()[-1500:])
            out['other_configs'][kind] = r
            if r.get('error') or (r.get('parity') and not r['parity']['ok']):
                bad.append(kind)
    emit(out)
    if world > 1:
        dist.destroy_process_group()
    if out.get('parity') and not out['parity']['ok']:
        raise SystemExit('parity check at the benchmarked configuration FAILED: ' + json.dumps(out['parity']))
    if bad:
        raise SystemExit(f'other_configs {bad}: error or parity failure (see the JSON line)')


def run_from_files(args, dev, synthetic_scans_per_s=None):
    """N4: the mv-3ddet train step FED FROM FILES -- a synthetic dataset in the EmbodiedScan layout (20+ JPEG and
    16-bit PNG frames of 480x640 per scan, the reference's .pkl annotations) in /dev/shm, EmbodiedScanDataset + the config's own
    train_pipeline, ScanLoader with forked persistent workers decoding into shared pinned slots (PointSample draws on the device),
    double-buffered H2D + resize on a copy stream, det.train_step.  Reports scans/s, the time the step loop waits for the loader,
    workers and granted cores.  configs/detection/mv-det3d_8xb4_embodiedscan-3d-284class-9dof.py:176-196, datasets/transforms/loading.py:53-81."""
    import shutil
    import tempfile
    import torch
    from embodiedscan_amd import engine as E, pipeline, synth
    from embodiedscan_amd.config import build_detector, build_optim_wrapper, load_config
    from embodiedscan_amd.datasets import EmbodiedScanDataset, ScanLoader
    from embodiedscan_amd.datasets.loader import effective_cpus
    E.PRECISION[0] = args.precision
    base = '/dev/shm' if os.path.isdir('/dev/shm') and os.access('/dev/shm', os.W_OK) else None
    root = tempfile.mkdtemp(prefix='es_files_', dir=base)
    steps, warmup = max(4, min(args.steps, args.other_steps)), 4
    # the step loop's own host-side tensor ops are tiny: one intra-op thread.  (After the CPU-oracle legs of the default run torch
    # holds a 16-thread OpenMP pool; with 12 decode workers on the 16 granted cores every small op then waits at an OpenMP barrier
    # behind descheduled threads: 30.5 ms / step against 26.1 alone, profiles/r5i_bench_default.json.)
    prev_threads = torch.get_num_threads()
    torch.set_num_threads(1)
    try:
        names = [f'class{i}' for i in range(284)]
        synth.write_dataset(root, n_scans=8, n_frames=22, height=480, width=640, n_boxes=25, class_names=names, seed=1,
                            n_voxels=(40, 40, 16), render_device=str(dev))
        cfg = load_config(os.path.join(ROOT, 'configs', 'mv_3ddet.py'))
        ds = EmbodiedScanDataset(root, 'embodiedscan_infos_train.pkl', metainfo=dict(classes=names), pipeline=cfg['train_pipeline'])
        cores = effective_cpus()
        # 3/4 of the granted cores decode (12 of 16 on the GPU boxes: 150.6 scans/s against 134.8 / 141.5 with 10 / 14, profiles/r5e_*): the
        # step loop's own host thread and the copy engine's completion handling need the rest of the quota
        workers = max(1, min(int(os.environ.get('ES_LOADER_WORKERS', max(1, int(cores) * 3 // 4))), int(cores)))
        ld = ScanLoader(ds, batch_size=args.batch, shuffle=True, seed=0, times=100000, num_threads=workers, prefetch=min(max(16, 2 * workers), 64),
                        pin=True, workers='process', exact_draws=False, device_draws=True)
        det = build_detector(cfg, device=dev, seed=0).to(dev)
        optim = build_optim_wrapper(cfg)
        it = iter(ld)
        copy = copy_stream(dev)
        slots = ready = done = None
        n, waits, losses = 0, [], None

        def step():
            nonlocal slots, ready, done, n, losses
            t0 = time.perf_counter()
            batch = next(it)                                       # blocks only if the workers are behind
            w = time.perf_counter() - t0
            if slots is None:
                slots = [[pipeline.alloc_slot(sc, dev) for sc in batch] for _ in range(2)]
                ready = [torch.cuda.Event() for _ in range(2)]
                done = [torch.cuda.Event() for _ in range(2)]
                for e in done:
                    e.record()
            k = n % 2
            with torch.cuda.stream(copy):
                copy.wait_event(done[k])                           # the step that read this device slot two steps ago is done
                dscans = [pipeline.upload_into(sl, sc) for sl, sc in zip(slots[k], batch)]
                ev = torch.cuda.Event()
                ev.record(copy)
                ready[k].record(copy)
            ld.done(batch, ev)                                     # the pinned slots are reusable once the copy has landed
            torch.cuda.current_stream().wait_event(ready[k])
            losses = det.train_step(pipeline.make_batch(dscans), optim)
            done[k].record()
            n += 1
            return w
        for _ in range(4 + warmup):
            step()
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
        ev[0].record()
        t0 = time.perf_counter()
        for i in range(steps):
            waits.append(step())
            ev[i + 1].record()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        it.close()
        ld.close()
        val = args.batch * steps / dt
        out = dict(metric='scans/sec (train step) mv-3ddet FED FROM FILES, 20x(480x640) RGB-D views', value=round(val, 3), unit='scans/s',
                   steps=steps, ms_per_step=round(dt / steps * 1e3, 3), scans_per_step=args.batch,
                   loader=dict(workers=workers, kind='forked processes, shared pinned slots, device-side PointSample draws',
                               cores_granted=cores, cpus_visible=os.cpu_count(),
                               wait_ms_per_step=dict(mean=round(sum(waits) / len(waits) * 1e3, 3), max=round(max(waits) * 1e3, 3)),
                               dataset=dict(scans=len(ds), frames_per_scan=22, where=root.rsplit('/', 1)[0])),
                   losses={k: round(float(v), 6) for k, v in losses.items()},
                   step_ms=[round(ev[i].elapsed_time(ev[i + 1]), 2) for i in range(steps)],
                   note='files in /dev/shm (page-cache speed: decode cost, not disk); JPEG decode = PIL on host cores (no rocJPEG in the image), 16-bit '
                        'PNG = csrc/host_codec.c; loader wait = host time next(loader) blocked per step')
        if synthetic_scans_per_s:
            out['vs_synthetic'] = round(val / synthetic_scans_per_s, 4)
            out['cores_for_8_gpus'] = f'{8 * workers} worker processes at this rate (one rank per GPU, {workers} each)'
        del det, optim
        E.TAPE.clear()
        release_frozen()
        return out
    finally:
        torch.set_num_threads(prev_threads)
        shutil.rmtree(root, ignore_errors=True)


OTHER = {
    'grounding': dict(cfg='mv_grounding.py', views=20, rotate=2, seed=777,
                      metric='scans/sec (train step) mv-grounding, 20x(480x640) RGB-D views',
                      workload='SparseFeatureFusion3DGrounder: ResNet-50(w16) + MinkResNet34 + MinkNeck + 6-layer decoder (256 queries, '
                               '8 heads, FFN 2048) + GroundingHead with device-side Hungarian; frozen random-init RoBERTa-base-shaped '
                               'text encoder; full train step incl. H2D of the batch and paramwise AdamW'),
    'occupancy': dict(cfg='mv_occ.py', views=10, rotate=3, seed=4321,
                      metric='scans/sec (train step) occupancy, 10x(480x640) RGB-D views, 40x40x16 volume',
                      workload='DenseFusionOccPredictor: ResNet-50 + FPN, MinkResNet34, IndoorImVoxelNeck 768-1536-3072, ImVoxelOccHead '
                               '81 classes, batch 1 (reference 8xb1), 751 M parameters; full train step incl. H2D of the batch and AdamW'),
}
ATTN = {'es_attn_fwd', 'es_attn_bwd'}


def run_other_config(kind, args, dev):
    """BASELINE config 4 (mv-grounding, `--grounding-batch` scans per step, reference 8xb12) or 5 (occupancy, batch 1) under
    the same rules as the primary line: host->device copy of every batch inside the step (one pinned slab, copy stream,
    double-buffered), rotating distinct batches, parity of the losses against the CPU oracle from the pre-training weights,
    roofline of the kernel family north_star names for it (attention MFMA / dense-neck MFMA), single-stream stage times."""
    import torch
    from embodiedscan_amd import engine as E, hip, pipeline
    from embodiedscan_amd.config import build_detector, build_optim_wrapper, load_config
    from embodiedscan_amd.synth import make_grounding_sample, make_occ_gt, make_scan
    o = OTHER[kind]
    hip.PAIRS.clear()                     # map pointers of the previous detector may be recycled
    batch = args.grounding_batch if kind == 'grounding' else 1
    steps, warmup = max(1, min(args.steps, args.other_steps)), max(5, min(args.warmup, 8))   # >= 5: the first steps of a fresh detector run
                                                                                             # 3-4x long (lazy maps, allocator, two graph captures)
    cfg = load_config(os.path.join(ROOT, 'configs', o['cfg']))
    det = build_detector(cfg, device=dev, seed=0).to(dev)
    optim = build_optim_wrapper(cfg)
    scans = []
    for i in range(batch * o['rotate']):
        sc = make_scan(o['seed'] + i, n_views=o['views'], augment=(kind == 'grounding'), render_device=str(dev))
        if kind == 'grounding':
            a = make_grounding_sample(sc, seed=i)
            sc = dict(sc, text=a['text'], tokens_positive=a['tokens_positive'], gt_boxes=a['gt_boxes'], gt_labels=a['gt_labels'])
        else:
            oc = make_occ_gt(sc, seed=i)
            sc = dict(sc, gt_occupancy=oc['gt_occupancy'], gt_occupancy_masks=oc['gt_occupancy_masks'])
        scans.append(sc)
    make = pipeline.make_grounding_batch if kind == 'grounding' else pipeline.make_occ_batch

    # ---- parity (untimed): losses of scans[0] alone, pre-training weights, HIP path vs the CPU oracle's forward
    parity = base = None
    if not args.no_cpu_baseline:
        parity, base = other_parity(kind, cfg, det, scans[0], make, dev, args)

    feeder = Feeder([pipeline.pin_batch(scans[r * batch:(r + 1) * batch]) for r in range(o['rotate'])], dev)

    def step():
        dscans = feeder.next()
        E.mark('_begin')
        out = det.train_step(make(dscans), optim)
        feeder.done()
        return out

    for _ in range(4):                         # priming (lazy maps, allocator, one hipGraph capture per preprocessor buffer), outside `warmup`
        step()
    for _ in range(warmup):
        losses = step()
    torch.cuda.synchronize()
    prof = {'names': ENGINE | ATTN | {'es_ground_match'}, 'records': [], 'event': lambda: torch.cuda.Event(enable_timing=True)}
    step_ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    diag = [] if os.environ.get('ES_BENCH_DIAG') else None      # per-step host / allocator / graph state (hunting step-time outliers)
    gc_log = []
    if diag is not None:
        import gc
        _t = [0.0]

        def _gc_cb(phase, info):                                 # every collection of the timed loop: generation and duration
            if phase == 'start':
                _t[0] = time.perf_counter()
            else:
                gc_log.append((len(diag), info['generation'], round((time.perf_counter() - _t[0]) * 1e3, 2), info['collected']))
        gc.callbacks.append(_gc_cb)
    step_ev[0].record()
    cfs0 = cfs_stat()
    t0 = time.perf_counter()
    for it in range(steps):
        h0 = time.perf_counter()
        losses = step()
        step_ev[it + 1].record()
        if diag is not None:
            ms = torch.cuda.memory_stats()
            diag.append(dict(host_ms=round((time.perf_counter() - h0) * 1e3, 2), reserved_MB=ms['reserved_bytes.all.current'] >> 20,
                             allocated_MB=ms['allocated_bytes.all.current'] >> 20, mallocs=ms['num_device_alloc'], frees=ms['num_device_free'],
                             retries=ms['num_alloc_retries'], graphs=dict(E.GRAPH_STATS),
                             extra=dict(getattr(det, 'diag', None) or {})))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    host = host_report(cfs0)
    hip.PROFILE = prof                          # one extra untimed step under the same schedule, every engine launch bracketed
    step()
    recs = resolve_pairs(hip, prof['records'])
    torch.cuda.synchronize()
    hip.PROFILE = None
    peak = K_PEAK_MFMA[args.precision]
    eng = engine_totals([r for r in recs if r[0] in ENGINE], peak)
    tok_timed = None
    if kind == 'grounding':                   # token counts of the batch the instrumented step ran on (the next extra step rotates on)
        tok_timed = (det.last_queries['klen'].cpu().tolist(), det.last_text['mask'].sum(1).cpu().tolist(), det.neck_3d.last['Lmax'])
    # one extra untimed step on the single-stream schedule: stage times and stand-alone launch durations
    saved = (E.TWO_STREAMS[0], E.WGRAD_ASYNC[0])
    E.TWO_STREAMS[0] = E.WGRAD_ASYNC[0] = False
    stages = stage_times(step)                  # median of 3 warmed single-stream steps
    prof1 = dict(prof, records=[])
    hip.PROFILE = prof1
    step()
    hip.PROFILE = None
    E.TWO_STREAMS[0], E.WGRAD_ASYNC[0] = saved
    r1 = resolve_pairs(hip, prof1['records'])
    torch.cuda.synchronize()
    stages['_note'] = 'single-stream schedule (ES_TWO_STREAMS=0 ES_WGRAD_ASYNC=0): HIP-event time between stage boundaries, MEDIAN of 3 untimed ' \
                      'steps after one warm-up step on that schedule'
    e1 = engine_totals([r for r in r1 if r[0] in ENGINE], peak)
    if os.environ.get('ES_BENCH_DUMP'):                         # dev: every engine launch of the single-stream step, in order
        with open(os.environ['ES_BENCH_DUMP'], 'w') as f:
            for name, ev0, ev1, a, _ in r1:
                if name in ENGINE:
                    nbr, n_out, n_in, K, cin, cout = engine_args(name, a)
                    f.write(json.dumps(dict(fn=name, K=K, cin=cin, cout=cout, n_out=n_out, n_in=n_in, map=bool(nbr),
                                            us=round(ev0.elapsed_time(ev1) * 1e3, 1),
                                            args=[x for x in a if isinstance(x, int) and abs(x) < (1 << 31)])) + '\n')

    traffic, traffic_dense, tnote = None, None, 'traffic: null (no PMC summary committed for this configuration)'
    for tag in ('r6', 'r5', 'r4', 'r3'):
        pmc_file = os.path.join(ROOT, 'profiles', f'{tag}_pmc_traffic_{kind}.json')
        if args.precision == 'bf16' and os.path.exists(pmc_file):
            pmc = json.load(open(pmc_file))
            traffic = pmc['bytes_per_launch']
            tnote = (f"traffic is STATIC: HBM bytes per launch of the named family from the committed PMC passes of `bench.py --only {kind}` "
                     f"(profiles/{tag}_pmc_traffic_{kind}.json: {pmc['launches'] // pmc['steps']} launches per step), not re-measured by this run")
            if 'dense' in pmc:                  # occupancy: the dense-volume kernels alone (k_dconv*: the neck)
                traffic_dense = pmc['dense']['bytes_per_launch']
                tnote = (f"traffic is STATIC: HBM bytes per k_dconv* kernel launch (slice reductions are launches of their own) from the "
                         f"committed PMC passes of `bench.py --only {kind}` (profiles/{tag}_pmc_traffic_{kind}.json: "
                         f"{pmc['dense']['launches'] // pmc['steps']} such launches per step), not re-measured by this run")
            break
    if kind == 'grounding':
        klen = det.last_queries['klen'].cpu().tolist()
        tl = det.last_text['mask'].sum(1).cpu().tolist()
        Lmax = det.neck_3d.last['Lmax']
        att = attention_totals(r1, klen, tl, Lmax)              # stand-alone durations (single-stream step)
        att_c = attention_totals(recs, *tok_timed)              # under the concurrent schedule (instrumented extra step)
        tfl = att['tflops']
        roofline = dict(bound='mfma', achieved=tfl, peak=peak, unit='TFLOP/s', frac=round(tfl / peak, 5), traffic=traffic,
                        kernel='attention: k_attn_fwd + k_attn_bwd_dq + k_attn_bwd_dkv (+ k_attn_delta), head_dim 32, 8 heads',
                        launches_per_step=att['launches'], kernel_ms_per_step=att['ms'], fwd=att['fwd'], bwd=att['bwd'],
                        concurrent_schedule=dict(kernel_ms_per_step=att_c['ms'], tflops=att_c['tflops']),
                        note='algorithmic flops per call: forward 4*H*Lq*Lk_valid*32, backward 10*H*Lq*Lk_valid*32 summed over the '
                             'samples (valid keys only); durations = HIP events on the launch stream, stand-alone (single-stream '
                             'step); ' + tnote)
        extra = dict(point_tokens=klen, text_tokens=tl,
                     hungarian_ms=round(sum(e0.elapsed_time(e1_) for n, e0, e1_, a, _ in r1 if n == 'es_ground_match'), 3))
    else:
        # the neck's launches: IndoorImVoxelNeck runs 768 / 1536 / 3072 channels (out blocks: -> 128), so one side is a multiple of 768.
        # Rounds 4 and 5a..m filtered on ">= 768 channels on a side", which also caught the image backbone's 1024- / 2048-channel
        # 1x1 launches (~2.7 ms of ~35 us row GEMMs per step, profiles/r5o_occ_launches.jsonl); that family is kept beside it.
        is_neck = lambda r: any(c % 768 == 0 for c in engine_args(r[0], r[3])[4:6])
        ge768 = lambda r: max(engine_args(r[0], r[3])[4:6]) >= 768
        nk = engine_totals([r for r in r1 if r[0] in ENGINE and is_neck(r)], peak)
        neck_c = engine_totals([r for r in recs if r[0] in ENGINE and is_neck(r)], peak)
        old = engine_totals([r for r in r1 if r[0] in ENGINE and ge768(r)], peak)
        if traffic_dense is not None:
            traffic = traffic_dense
        roofline = dict(bound='mfma', achieved=nk['tflops'], peak=peak, unit='TFLOP/s', frac=round(nk['tflops'] / peak, 4),
                        traffic=traffic,
                        kernel='convolution engine on the dense 3-D neck (IndoorImVoxelNeck: every engine launch with 768 / 1536 / 3072 '
                               'channels on a side)',
                        launches_per_step=nk['launches'], kernel_ms_per_step=nk['ms'], frac_of_binding_roof=nk['frac_binding'],
                        concurrent_schedule=dict(kernel_ms_per_step=neck_c['ms'], tflops=neck_c['tflops']),
                        family_ge768=dict(launches_per_step=old['launches'], kernel_ms_per_step=old['ms'], tflops=old['tflops'],
                                          frac=round(old['tflops'] / peak, 4),
                                          note='the filter of rounds 4 / 5a..m (>= 768 channels on a side): the neck PLUS the image '
                                               "backbone's 1024- / 2048-channel 1x1 launches; comparable with VERDICT r4's 0.129"),
                        note='algorithmic flops = 2*P*Cin*Cout per launch (P = valid (output, tap) pairs of the dense 3x3x3 maps); '
                             'durations = HIP events on the launch stream, stand-alone (single-stream step); ' + tnote)
        extra = {}
    res = dict(metric=o['metric'], value=round(batch * steps / dt, 4), unit='scans/s', n_gpus=1, steps=steps, warmup=warmup,
               ms_per_step=round(dt / steps * 1e3, 3), higher_is_better=True, dtype=args.precision, data='synthetic',
               config=dict(workload=o['workload'], scans_per_gpu_per_step=batch, views=o['views'], distinct_batches=o['rotate'],
                           h2d=f'{feeder.h2d_bytes / 1e6:.0f} MB per step as ONE copy from a pinned slab on a copy stream, double-buffered'),
               losses={k: round(float(v), 6) for k, v in losses.items()}, roofline=roofline,
               engine_all=dict(launches_per_step=eng['launches'], kernel_ms_per_step=eng['ms'], tflops=eng['tflops'],
                               frac_of_binding_roof=eng['frac_binding'], compulsory_GBps=eng['comp_GBps'],
                               single_stream=dict(kernel_ms_per_step=e1['ms'], tflops=e1['tflops'], frac_of_binding_roof=e1['frac_binding'])),
               classes=launch_classes(r1, peak, top=16), stage_ms=stages, optimizer_pass=OPT_PASS.get(optim.last_path, optim.last_path), host=host,
               step_ms=[round(step_ev[i].elapsed_time(step_ev[i + 1]), 2) for i in range(steps)], **extra)
    if diag is not None:
        import gc
        gc.callbacks.remove(_gc_cb)
        res['step_diag'] = diag
        res['gc_collections'] = dict(count=len(gc_log), gen2=[g for g in gc_log if g[1] == 2], slow=[g for g in gc_log if g[2] >= 2.0],
                                     note='(step index, generation, ms, objects collected) of the collections inside the timed loop')
    if parity is not None:
        res['parity'], res['cpu_baseline'] = parity, base
    del det, optim, feeder
    E.TAPE.clear()
    release_frozen()
    return res


def attention_totals(records, klen, tl, Lmax):
    att = dict(fwd=[0.0, 0.0, 0], bwd=[0.0, 0.0, 0])
    for name, e0, e1, a, _ in records:
        if name not in ATTN:
            continue
        t = e0.elapsed_time(e1)
        if name == 'es_attn_fwd':
            Bn, H, Lq, Lk, kl = a[6], a[7], a[8], a[9], a[10]
        else:
            Bn, H, Lq, Lk, kl = a[11], a[12], a[13], a[14], a[15]
        valid = sum(klen) if (kl and Lk == Lmax) else (sum(tl) if kl else Bn * Lk)
        d = att['fwd' if name == 'es_attn_fwd' else 'bwd']
        d[0] += t
        d[1] += (4.0 if name == 'es_attn_fwd' else 10.0) * H * Lq * valid * 32
        d[2] += 1
    ms, fl = att['fwd'][0] + att['bwd'][0], att['fwd'][1] + att['bwd'][1]
    one = lambda d: dict(ms=round(d[0], 3), launches=d[2], tflops=round(d[1] / max(d[0], 1e-9) / 1e9, 3))
    return dict(ms=round(ms, 3), launches=att['fwd'][2] + att['bwd'][2], tflops=round(fl / max(ms, 1e-9) / 1e9, 3),
                fwd=one(att['fwd']), bwd=one(att['bwd']))


def cfs_stat(root='/sys/fs/cgroup'):
    """(periods in which the kernel throttled this cgroup, microseconds throttled) so far, or None without a cgroup-v2 cpu.stat"""
    try:
        d = dict(l.split() for l in open(os.path.join(root, 'cpu.stat')).read().splitlines())
        return int(d['nr_throttled']), int(d['throttled_usec'])
    except (OSError, KeyError, ValueError):
        return None


def host_report(before):
    """host side of a timed region: torch's intra-op pool, the cores the cgroup grants, and how often the kernel's CPU-bandwidth
    controller stopped the process inside the region (a throttled process cannot queue GPU work: profiles/r5w_*)"""
    import torch
    from embodiedscan_amd.datasets.loader import effective_cpus
    after = cfs_stat()
    out = dict(torch_threads=torch.get_num_threads(), granted_cores=effective_cpus(), visible_cpus=os.cpu_count())
    if before is not None and after is not None:
        out.update(cfs_throttled_periods=after[0] - before[0], cfs_throttled_ms=round((after[1] - before[1]) / 1e3, 1))
    return out


OPT_PASS = {'table': 'es_adamw_table: clip + AdamW + the bf16 copies of the kernels in one pass',
            'flat': 'es_adamw_step (+ es_cast_weights_table at the next step)'}


def launch_classes(records, mfma_peak, top=6):
    """the engine launches of one step grouped into launch classes (entry point x tap count x channel widths): launches, ms,
    algorithmic TFLOP, compulsory GB and the fraction of the binding roof of each -- the dominant kernels' fractions
    without arithmetic"""
    groups = {}
    for r in records:
        name, a = r[0], r[3]
        if name not in ENGINE:
            continue
        nbr, n_out, n_in, K, cin, cout = engine_args(name, a)
        kind = 'wgrad' if (name.startswith('es_spconv_wgrad') or name in DENSE_WGRAD or name in IMGW or name in ROWW) else 'fwd/dgrad'
        if name in DENSE:
            kind += ' dense'
        key = f'{kind} K={K} {cin}->{cout}'
        groups.setdefault(key, []).append(r)
    rows = []
    for key, rs in groups.items():
        t = engine_totals(rs, mfma_peak)
        rows.append(dict(cls=key, launches=t['launches'], ms=t['ms'], tflop=round(t['tflops'] * t['ms'] * 1e-3, 4),
                         compulsory_GB=round(t['comp_bytes'] / 1e9, 3), tflops=t['tflops'], compulsory_GBps=t['comp_GBps'],
                         frac_of_binding_roof=t['frac_binding']))
    rows.sort(key=lambda d: -d['ms'])
    return rows[:top]


def _cap_cpu_threads():
    """the CPU legs use as many threads as the box grants cores (cgroup quota), not as many as it shows logical CPUs: on the
    GPU boxes (256 shown, 16 granted) torch's default 128 threads time-slice 16 cores; `cores` in the line is this count"""
    import torch
    from embodiedscan_amd.datasets.loader import effective_cpus
    n = effective_cpus()                    # (set explicitly: the train steps run with the engine's smaller pool, settle_host_threads)
    if n != torch.get_num_threads():
        torch.set_num_threads(n)
    return n


def _gpu_leg_threads():
    """back to the train steps' host pool after a CPU leg"""
    from embodiedscan_amd import engine as E
    E.settle_host_threads(force=True)


def other_parity(kind, cfg, det, scan, make, dev, args):
    """losses of ONE scan from the pre-training weights: HIP path vs the CPU oracle's forward (no gradients; the oracle's
    forward+backward is what the primary line's cpu_baseline times).  Tolerance: bf16 5e-2 (grounding: 12 losses over 6
    decoder layers; occupancy: CE + sem_scal + geo_scal per level, the finest level measured at 2.9e-2 at random init),
    exact-f32 mode 1e-3."""
    import torch
    from embodiedscan_amd import engine as E, pipeline
    from oracle import model as OM
    sd0 = {k: v.cpu() for k, v in det.state_dict().items()}
    d0 = pipeline.upload_scan(scan, dev)
    d0.update({k: scan[k] for k in ('text', 'tokens_positive', 'gt_occupancy', 'gt_occupancy_masks') if k in scan})
    b0 = make([d0])
    points = [p.cpu() for p in b0['inputs']['points']]
    E.TAPE.clear()
    data = det.data_preprocessor(b0, True)
    det._bind()
    l0 = det.forward(data['inputs'], data['data_samples'], mode='loss')
    torch.cuda.synchronize()
    hipl = {k: float(v) for k, v in l0.items()}
    E.TAPE.clear()
    E.join_wgrad_streams()
    imgs = OM.preprocess_img(torch.from_numpy(scan['img']), [123.675, 116.28, 103.53], [58.395, 57.12, 57.375])[None]
    _cap_cpu_threads()
    t0 = time.perf_counter()
    with torch.no_grad():
        if kind == 'grounding':
            from oracle import grounding as OG
            th, tm = det.last_text['hidden'].float().cpu(), det.last_text['mask'].cpu()
            pms = [ds.gt_instances_3d.positive_maps.cpu() for ds in data['data_samples']]
            ol = OG.grounder_loss(sd0, points, imgs, [scan['meta']], th, tm, [torch.as_tensor(scan['gt_boxes'])], pms,
                                  num_queries=det.num_queries, num_layers=det.decoder.num_layers,
                                  thr=cfg['model']['neck_3d']['pts_prune_threshold'])
            tol = 5e-2
        else:
            from oracle import occ as OO
            m = cfg['model']
            ol = OO.detector_loss(sd0, points, imgs, [scan['meta']], [torch.from_numpy(scan['gt_occupancy'])],
                                  [torch.from_numpy(scan['gt_occupancy_masks'])], m['n_voxels'], m['point_cloud_range'],
                                  cfg['prior_generator']['ranges'][0], tuple(m['neck_3d']['n_blocks']))
            tol = 5e-2
    dt = time.perf_counter() - t0
    if args.precision != 'bf16':
        tol = 1e-3
    rel = {k: abs(hipl[k] - float(ol[k])) / max(abs(float(ol[k])), 1e-6) for k in ol}
    parity = dict(what=f'{len(ol)} losses of scans[0] alone, pre-training weights, HIP path vs CPU oracle (f32)',
                  hip={k: round(v, 6) for k, v in hipl.items()}, oracle={k: round(float(v), 6) for k, v in ol.items()},
                  rel_err={k: float(f'{v:.3e}') for k, v in rel.items()}, tol=tol, ok=bool(max(rel.values()) < tol))
    base = dict(value=round(1.0 / dt, 5), unit='scans/s', cores=torch.get_num_threads(), kind='port',
                sample=f'1 scan x {scan["depth"].shape[0]} views 480x640, 100k points, ONE FORWARD (no backward, no optimiser) of the '
                       f'PyTorch-f32 CPU oracle, {dt:.1f} s')
    del d0, b0, data, l0
    _gpu_leg_threads()
    return parity, base


def dense_info(name, a):
    """(n_out, n_in, K, cin, cout, valid (output, tap) pairs) of a dense-engine launch, in the convention of the map launches
    (the data gradient is a forward launch over the input voxels with the channel roles swapped)"""
    fwd = name == 'es_dconv_fwd_bf16'
    mode = a[4] if fwd else (5 if a[5] else 2)
    geom = list(a[3] if fwd else a[4])
    B, X, Y, Z, ks, st, pad = geom
    cin, cout = (a[5], a[6]) if fwd else (a[6], a[7])
    o = lambda d: (d + 2 * pad - ks) // st + 1
    ax = lambda d: sum(1 for q in range(o(d)) for k in range(ks) if 0 <= q * st - pad + k < d)
    if Z == 0:                                   # flat grid: nn.Conv2d on (B, X, Y) images, ks x ks taps
        pairs, n_in, n_out, K = float(B) * ax(X) * ax(Y), B * X * Y, B * o(X) * o(Y), ks ** 2
    else:
        pairs, n_in, n_out, K = float(B) * ax(X) * ax(Y) * ax(Z), B * X * Y * Z, B * o(X) * o(Y) * o(Z), ks ** 3
    if mode >= 3:
        # nn.ConvTranspose3d(k = 2, s = 2): 8 taps, every (coarse voxel, tap) pair is valid; forward / weight gradient read Cin and
        # write / pair with Cout on the fine grid, the data gradient the other way round
        n_c, n_f, pairs = B * X * Y * Z, 8 * B * X * Y * Z, 8.0 * B * X * Y * Z
        if mode == 4:
            return n_c, n_f, 8, cout, cin, pairs
        return n_f, n_c, 8, cin, cout, pairs
    if mode == 1:
        return n_in, n_out, K, cout, cin, pairs
    return n_out, n_in, K, cin, cout, pairs


def resolve_pairs(hip, records):
    """map pointer -> pair counter while the maps are still alive"""
    out = []
    for name, e0, e1, a in records:
        key = None
        if name in DENSE:
            out.append((name, e0, e1, a, dense_info(name, a)[5]))
            continue
        if name in ROWW:
            out.append((name, e0, e1, a, float(a[4])))
            continue
        if name in IMGC:
            axc = lambda d: sum(1 for q in range(d // a[7]) for k in range(3) if 0 <= q * a[7] - 1 + k < d)
            out.append((name, e0, e1, a, float(a[3]) * axc(a[4]) * axc(a[5])))
            continue
        if name in IMGW:                                 # valid (pixel, tap) pairs of a 3x3 / pad 1 / stride 1 image convolution
            ax = lambda d: sum(1 for q in range(d // a[8]) for k in range(3) if 0 <= q * a[8] - 1 + k < d)
            out.append((name, e0, e1, a, float(a[4]) * ax(a[5]) * ax(a[6])))
            continue
        if name == 'es_spconv_wgrad_bf16_src':
            key = a[6]
        elif name in HALO:
            key = a[3]                                   # (the plan's position table stands for the map)
        elif name in ENGINE:
            key = a[4] if (name.startswith('es_spconv_wgrad') or name in FWD_X) else a[3]
        out.append((name, e0, e1, a, hip.PAIRS.get(key)))
    return out


def engine_args(name, a):
    if name in DENSE:
        n_out, n_in, K, cin, cout, _ = dense_info(name, a)
        return 1, n_out, n_in, K, cin, cout
    if name == 'es_spconv_wgrad_bf16_src':      # (X, x_half, ldx, dY, dy_half, ldy, nbr, n_out, n_in, K, Cin, Cout, dW, stream)
        return a[6], a[7], a[8], a[9], a[10], a[11]
    if name in HALO:
        return a[3], a[6], a[7], a[8], a[9], a[10]
    if name in IMGW:
        return 1, a[4] * (a[5] // a[8]) * (a[6] // a[8]), a[4] * a[5] * a[6], 9, a[7], a[7]
    if name in ROWW:
        return 0, a[4], a[4], 1, a[5], a[6]
    if name in IMGC:
        return 1, a[3] * (a[4] // a[7]) * (a[5] // a[7]), a[3] * a[4] * a[5], 9, a[6], a[6]
    if name in FWD_X:
        return a[4], a[5], a[6], a[7], a[8], a[9]
    if not name.startswith('es_spconv_wgrad'):
        return a[3], a[4], a[5], a[6], a[7], a[8]
    return a[4], a[5], a[6], a[7], a[8], a[9]


def engine_totals(records, mfma_peak):
    """sums over the convolution-engine launches of one step: HIP-event ms, algorithmic flops, SURVEY-8(d) pair bytes,
    compulsory bytes, and the per-launch binding roof time"""
    ms = flop = pair_b = comp_b = t_bind = t_mfma = t_hbm = 0.0
    n = 0
    for name, e0, e1, a, pairs_dev in records:
        ms += e0.elapsed_time(e1)
        n += 1
        nbr, n_out, n_in, K, cin, cout = engine_args(name, a)
        if isinstance(pairs_dev, float):
            pairs = pairs_dev
        else:
            pairs = float(pairs_dev.item()) if pairs_dev is not None else (float(min(n_out, n_in)) if not nbr else float(n_out) * K)
        wgrad = name.startswith('es_spconv_wgrad') or name in DENSE_WGRAD or name in IMGW or name in ROWW
        wb = 2 if ('bf16' in name and not wgrad) else 4
        f = 2.0 * pairs * cin * cout
        pb = pairs * (cin + cout) * 4.0 + float(K) * cin * cout * wb
        # compulsory: fwd/dgrad read n_in rows of Cin, write n_out rows of Cout, read the weights once;
        # wgrad reads both row matrices once and read-modify-writes the f32 weight gradient
        # (rows gathered from a bf16 shadow count 2 B per element, the shadow's own cast pass is not an engine launch)
        bx = by = 4.0
        if name == 'es_spconv_wgrad_bf16_src':
            bx, by = (2.0 if a[1] else 4.0), (2.0 if a[4] else 4.0)
        elif name in IMGC:
            bx, by = (2.0, 2.0 if a[15] else 4.0) if a[8] == 0 else (4.0, 4.0)     # forward: bf16 in, bf16 / f32 out; data gradient: f32 in / out
        elif name in IMGW or name in ROWW:
            bx = 2.0                                     # bf16 activation rows, f32 gradient rows
        elif name in DENSE_WGRAD:
            bx = by = 2.0
        elif name in DENSE or name in HALO or (name in FWD_X and a[1]):
            bx = 2.0
        if name == 'es_spconv_fwd_bf16_io' and a[17]:           # bf16 activation rows written by the image backbone
            by = 2.0
        cb = float(n_in) * cin * bx + float(n_out) * cout * by + float(K) * cin * cout * (8.0 if wgrad else wb)
        tm, th = f / (mfma_peak * 1e12), cb / (K_PEAK_HBM * 1e9)
        flop, pair_b, comp_b = flop + f, pair_b + pb, comp_b + cb
        t_mfma, t_hbm, t_bind = t_mfma + tm, t_hbm + th, t_bind + max(tm, th)
    s = ms * 1e-3
    return dict(ms=round(ms, 3), launches=n, tflops=round(flop / s / 1e12, 2) if s else 0.0,
                pair_GBps=round(pair_b / s / 1e9, 1) if s else 0.0, comp_GBps=round(comp_b / s / 1e9, 1) if s else 0.0,
                pair_bytes=pair_b, comp_bytes=comp_b, t_mfma=t_mfma, t_hbm=t_hbm,
                frac_binding=round(t_bind / s, 4) if s else 0.0)


def scatter_totals(records):
    """achieved HBM GB/s of the scatter path (A1-A4, A6, A8): algorithmic bytes of each launch from its arguments"""
    per = {}
    for name, e0, e1, a, _ in records:
        t = e0.elapsed_time(e1)
        if name == 'es_voxel_keys':            # (points, n, ld, batch, vs, keys)
            b = a[1] * (12 + 8)
        elif name == 'es_unique_first':        # (keys, n, tkeys, tvals, cap, ...): read keys, insert (key,row), emit (key,src)
            b = a[1] * (8 + 12 + 12)
        elif name == 'es_morton_sort':         # (keys, src, n, ...): 64-bit key + payload through the radix passes once
            b = a[2] * (12 + 12)
        elif name == 'es_stride_keys':
            b = a[1] * 16
        elif name == 'es_kernel_map':          # (out_keys, n_out, tk, tv, cap, ksize, in_ts, nbr): K probes of 12 B + 4 B out
            b = a[1] * (8 + a[5] ** 3 * (12 + 4))
        elif name == 'es_inverse_map':         # (nbr, n_out, K, n_in, inv)
            b = (a[1] + a[3]) * a[2] * 4
        elif name == 'es_union_plan':          # (ka, na, tk, tv, cap, kb, nb, ...)
            b = (a[1] + a[6]) * (8 + 12 + 4 + 8)
        elif name == 'es_point_sample_fwd':    # (coords, n, vs, meta, ms, V, feats, Hf, Wf, C, out, ldo, pix, cnt)
            b = a[1] * (16 + a[5] * a[9] * 4 + a[9] * 4 + a[5] * 4 + 4)
        elif name == 'es_point_sample_bwd':    # (coords, n, V, dout, ldo, pix, cnt, Hf, Wf, C, dfeats, n_img, head, next, acc)
            npix = a[11] * a[7] * a[8]             # compulsory: pix + link words, one read of every dout row, every pixel written
            b = a[1] * (a[2] * 8 + 4 + a[9] * 4) + npix * (4 + a[9] * 4)
        elif name == 'es_depth_to_points':     # (depth, H, W, sel_view, sel_pix, n, ...)
            b = a[5] * (4 + 4 + 4 + 12)
        else:
            continue
        d = per.setdefault(name, [0.0, 0.0, 0])
        d[0] += t
        d[1] += b
        d[2] += 1
    tot_ms = sum(v[0] for v in per.values())
    tot_b = sum(v[1] for v in per.values())
    return dict(achieved_GBps=round(tot_b / (tot_ms * 1e-3) / 1e9, 1) if tot_ms else 0.0, peak_GBps=K_PEAK_HBM,
                frac=round(tot_b / (tot_ms * 1e-3) / 1e9 / K_PEAK_HBM, 4) if tot_ms else 0.0, ms_per_step=round(tot_ms, 3),
                per_kernel={k: dict(ms=round(v[0], 3), launches=v[2], GBps=round(v[1] / (v[0] * 1e-3) / 1e9, 1) if v[0] else 0.0)
                            for k, v in sorted(per.items())},
                note='A1-A4/A6/A8 kernels of one untimed single-stream step; algorithmic bytes per launch (keys, table '
                     'probes of 12 B, map entries, gathered feature vectors), HIP events on the launch stream')


def cpu_baseline(scan, sd0, det, parity_hip, args):
    """The CPU oracle (a restatement, kind='port') timed on this box's host cores on ONE scan of the same workload:
    forward + backward of the detector loss (no optimiser), from the PRE-training weights.  Bounded sample, reported
    next to the GPU number.  The same run is the parity check at the benchmarked configuration: its three losses
    against the HIP path's on the same scan / weights (tolerance 2e-2 in bf16 mode, 1e-3 in f32 mode)."""
    import torch
    from oracle import model as OM, pipeline as OP
    names = set(det.arena.grad_dict().keys())
    sd = {k: v.clone().requires_grad_(k in names) for k, v in sd0.items()}
    _cap_cpu_threads()
    t0 = time.perf_counter()
    pts = [OP.scan_to_points(scan)]
    imgs = OM.preprocess_img(torch.from_numpy(scan['img']), [123.675, 116.28, 103.53], [58.395, 57.12, 57.375])[None]
    losses = OM.detector_loss(sd, pts, imgs, [scan['meta']], [torch.from_numpy(scan['gt_boxes'])],
                              [torch.from_numpy(scan['gt_labels'])])
    sum(losses.values()).backward()
    dt = time.perf_counter() - t0
    tol = 2e-2 if args.precision == 'bf16' else 1e-3
    rel = {k: abs(parity_hip[k] - float(losses[k])) / abs(float(losses[k])) for k in losses}
    parity = dict(what='three losses of scans[0] (20 views, 100k points), pre-training weights, HIP path vs CPU oracle (f32)',
                  hip={k: round(v, 6) for k, v in parity_hip.items()}, oracle={k: round(float(v), 6) for k, v in losses.items()},
                  rel_err={k: float(f'{v:.3e}') for k, v in rel.items()}, tol=tol, ok=bool(max(rel.values()) < tol))
    base = dict(value=round(1.0 / dt, 5), unit='scans/s', cores=torch.get_num_threads(), kind='port',
                sample=f'1 scan x {scan["depth"].shape[0]} views 480x640, 100k points, one forward+backward of the '
                       f'PyTorch-f32 CPU oracle (no optimiser step), {dt:.1f} s; os.cpu_count()={os.cpu_count()}')
    _gpu_leg_threads()
    return base, parity


if __name__ == '__main__':
    main()
