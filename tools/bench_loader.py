"""Host side of the real-data path: how many scans/s the threaded loader decodes (20 JPEG + 20 16-bit PNG of
480x640 per scan, the view / pixel / augmentation draws, pinned hand-over), per thread count, and with the copy + device
resize + A1-A3 attached.  Files live in /dev/shm (page-cache speed: the codec cost is what is measured).
    python tools/bench_loader.py [--scans 8] [--threads 1,4,8,16,32,64]"""
import argparse
import json
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--scans', type=int, default=8)
    ap.add_argument('--frames', type=int, default=24)
    ap.add_argument('--threads', default='1,4,8,16,32,64')
    ap.add_argument('--repeat', type=int, default=200)
    ap.add_argument('--seconds', type=float, default=8.0)
    ap.add_argument('--device-workers', type=int, default=64)
    ap.add_argument('--fast-draws', action='store_true', help='O(k) PointSample draws instead of the exact RandomState stream')
    ap.add_argument('--device-draws', action='store_true', help='the host only decodes; both PointSample draws run on the GPU (round 4)')
    ap.add_argument('--kinds', default='thread,process')
    ap.add_argument('--no-device', action='store_true', help='decode legs only (skip the copy + device resize + A1-A3 leg)')
    args = ap.parse_args()
    import torch
    from embodiedscan_amd import pipeline, synth
    from embodiedscan_amd.config import load_config
    from embodiedscan_amd.datasets import EmbodiedScanDataset, ScanLoader
    dev = torch.device('cuda:0') if torch.cuda.is_available() else None
    base = '/dev/shm' if os.path.isdir('/dev/shm') else None
    root = tempfile.mkdtemp(prefix='es_loader_', dir=base)
    try:
        t0 = time.time()
        names = [f'class{i}' for i in range(284)]
        synth.write_dataset(root, n_scans=args.scans, n_frames=args.frames, height=480, width=640, n_boxes=25,
                            class_names=names, seed=1, n_voxels=(40, 40, 16), render_device='cuda' if dev else 'cpu')
        n_files = sum(len(f) for _, _, f in os.walk(root))
        size = sum(os.path.getsize(os.path.join(d, f)) for d, _, fs in os.walk(root) for f in fs)
        cfg = load_config(os.path.join(ROOT, 'configs', 'mv_3ddet.py'))
        ds = EmbodiedScanDataset(root, 'embodiedscan_infos_train.pkl', metainfo=dict(classes=names),
                                 pipeline=cfg['train_pipeline'])
        out = dict(dataset=dict(scans=len(ds), frames_per_scan=args.frames, files=n_files, MB=round(size / 1e6, 1),
                                write_s=round(time.time() - t0, 1)),
                   pipeline=dict(n_images=ds.pipeline.n_images, n_points=ds.pipeline.n_points, img_scale=ds.pipeline.img_scale),
                   host_cores=os.cpu_count(), draws='device' if args.device_draws else ('host O(k)' if args.fast_draws else 'host exact stream'),
                   decode=[])
        for kind in args.kinds.split(','):
            for th in [int(t) for t in args.threads.split(',')]:
                if kind == 'thread' and th > 16:
                    continue                                  # GIL-bound: more threads do not help (see loader.py)
                ld = ScanLoader(ds, batch_size=4, shuffle=True, seed=0, times=args.repeat, num_threads=th,
                                prefetch=min(max(16, 2 * th), 64), pin=dev is not None, workers=kind,
                                exact_draws=not args.fast_draws, device_draws=args.device_draws)
                it = iter(ld)
                ld.done(next(it))                             # untimed: forks the workers, allocates and pins the slots
                t = time.time()
                n = 0
                for b in it:
                    n += len(b)
                    ld.done(b)
                    if time.time() - t >= args.seconds:
                        break
                dt = time.time() - t
                it.close()
                ld.close()
                out['decode'].append(dict(workers=kind, n=th, scans=n, scans_per_s=round(n / dt, 2),
                                          ms_per_scan=round(dt / n * 1e3, 1), pinned=bool(b[0]['depth'].is_pinned())))
                print(out['decode'][-1], file=sys.stderr)
        if dev is not None and not args.no_device:
            # loader -> copy stream (H2D + resize) -> A1-A3 on the compute stream, double-buffered like bench.py
            th = args.device_workers
            ld = ScanLoader(ds, batch_size=4, shuffle=True, seed=0, times=args.repeat * 2, num_threads=th,
                            prefetch=min(2 * th, 64), pin=True, workers='process', exact_draws=not args.fast_draws,
                            device_draws=args.device_draws)
            it = iter(ld)
            ld.done(next(it))
            copy = torch.cuda.Stream()
            slots, ready, n = None, None, 0
            torch.cuda.synchronize()
            t = time.time()
            for batch in it:
                if slots is None:
                    slots = [[pipeline.alloc_slot(s, dev) for s in batch] for _ in range(2)]
                    ready = [torch.cuda.Event() for _ in range(2)]
                    done = [torch.cuda.Event() for _ in range(2)]
                    for e in done:
                        e.record()
                k = (n // 4) % 2
                with torch.cuda.stream(copy):
                    copy.wait_event(done[k])
                    dscans = [pipeline.upload_into(sl, s) for sl, s in zip(slots[k], batch)]
                    ev = torch.cuda.Event()
                    ev.record(copy)
                    ready[k].record(copy)
                ld.done(batch, ev)                            # pinned slots reusable once the copy has landed
                torch.cuda.current_stream().wait_event(ready[k])
                data = pipeline.make_batch(dscans)
                done[k].record()
                n += len(batch)
                if time.time() - t >= args.seconds:
                    break
            torch.cuda.synchronize()
            dt = time.time() - t
            it.close()
            ld.close()
            out['to_device'] = dict(workers='process', n=th, scans=n, scans_per_s=round(n / dt, 2),
                                    h2d_MB_per_scan=round(pipeline.scan_h2d_bytes(batch[0]) / 1e6, 2),
                                    note='decode + pinned hand-over + async H2D + es_resize_u8 + es_depth_to_points')
        print(json.dumps(out))
    finally:
        shutil.rmtree(root, ignore_errors=True)


if __name__ == '__main__':
    main()
