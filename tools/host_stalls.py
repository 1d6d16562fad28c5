"""Where does the host thread lose time in a slow step?  (round 5: every ~9th occupancy step is 12 - 20 ms longer, the device idle at
the step's first host synchronisation -- profiles/r5t_occ_outlier.txt.)
    python tools/host_stalls.py occupancy|mv3ddet|grounding [steps] [threshold_ms]
wraps every entry point of libes_hip.so (and sparse.read_ints, the count read-back) with host timers and prints, per step, the wall
time and every C call or every stretch of Python BETWEEN two C calls that took longer than the threshold."""
import gc
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else 'occupancy'
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    thr = float(sys.argv[3]) if len(sys.argv) > 3 else 2.0
    import torch
    import bench
    from embodiedscan_amd import engine as E, hip, pipeline, sparse
    from embodiedscan_amd.config import build_detector, build_optim_wrapper, load_config
    from embodiedscan_amd.synth import make_grounding_sample, make_occ_gt, make_scan
    dev = torch.device('cuda:0')
    E.PRECISION[0] = 'bf16'
    cfgname = {'grounding': 'mv_grounding.py', 'occupancy': 'mv_occ.py', 'mv3ddet': 'mv_3ddet.py'}[kind]
    cfg = load_config(os.path.join(ROOT, 'configs', cfgname))
    det = build_detector(cfg, device=dev, seed=0).to(dev)
    optim = build_optim_wrapper(cfg)
    nv = 10 if kind == 'occupancy' else 20
    nscan = {'mv3ddet': 4, 'grounding': 12, 'occupancy': 1}[kind]
    batches = []
    for b in range(3):
        scans = []
        for i in range(nscan):
            sc = make_scan(100 + 10 * b + i, n_views=nv, augment=(kind == 'grounding'), render_device=str(dev))
            if kind == 'grounding':
                a = make_grounding_sample(sc, seed=i)
                sc = dict(sc, text=a['text'], tokens_positive=a['tokens_positive'], gt_boxes=a['gt_boxes'], gt_labels=a['gt_labels'])
            elif kind == 'occupancy':
                oc = make_occ_gt(sc, seed=i)
                sc = dict(sc, gt_occupancy=oc['gt_occupancy'], gt_occupancy_masks=oc['gt_occupancy_masks'])
            scans.append(sc)
        batches.append(pipeline.pin_batch(scans))
    make = {'grounding': pipeline.make_grounding_batch, 'occupancy': pipeline.make_occ_batch, 'mv3ddet': pipeline.make_batch}[kind]
    if os.environ.get('HS_THREADS1') == '1':
        torch.set_num_threads(1)
    feeder = bench.Feeder(batches, dev, resident=os.environ.get('HS_RESIDENT') == '1')
    print(f'torch intra-op threads {torch.get_num_threads()}, feeder resident {feeder.resident}')

    def step():
        out = det.train_step(make(feeder.next()), optim)
        feeder.done()
        return out
    for _ in range(6):
        step()
    torch.cuda.synchronize()
    log, last = [], [time.perf_counter(), 'start']
    cur = [0]

    def wrap(name, f):
        def g(*a):
            t0 = time.perf_counter()
            if (t0 - last[0]) * 1e3 > thr:
                log.append((cur[0], 'python before', name, round((t0 - last[0]) * 1e3, 2), 'after ' + last[1]))
            r = f(*a)
            t1 = time.perf_counter()
            if (t1 - t0) * 1e3 > thr:
                log.append((cur[0], 'inside', name, round((t1 - t0) * 1e3, 2), ''))
            last[0], last[1] = t1, name
            return r
        return g
    # watchdog (HS_WATCH=1): while one C call of the main thread has been running for more than HS_WATCH_MS, sample the kernel's view of
    # every thread of the process each millisecond: state, wait channel, CPU ticks -- what is the host doing while the device idles?
    import threading
    watch = os.environ.get('HS_WATCH') == '1'
    watch_ms = float(os.environ.get('HS_WATCH_MS', '26'))
    main_tid = threading.get_native_id()
    inflight = [None]                       # (name, t0) of the C call in flight
    samples = []                            # (step, call, ms into the call, {tid: (comm, state, wchan, utime + stime)})
    stop = [False]

    def rd(path):
        try:
            return open(path).read().strip()
        except Exception as e:
            return f'<{type(e).__name__}>'

    def watchdog():
        while not stop[0]:
            time.sleep(0.001)
            c = inflight[0]
            if c is None:
                continue
            el = (time.perf_counter() - c[1]) * 1e3
            if el < watch_ms:
                continue
            snap = {}
            for tid in os.listdir('/proc/self/task'):
                st = rd(f'/proc/self/task/{tid}/stat')
                try:
                    comm = st[st.index('(') + 1:st.rindex(')')]
                    f = st[st.rindex(')') + 2:].split()
                    snap[int(tid)] = (comm, f[0], rd(f'/proc/self/task/{tid}/wchan'), int(f[11]) + int(f[12]))
                except Exception:
                    pass
            stack = rd(f'/proc/self/task/{main_tid}/stack') if len(samples) % 4 == 0 else ''
            samples.append((cur[0], c[0], round(el, 1), snap, stack))
    if watch:
        threading.Thread(target=watchdog, daemon=True).start()
    _wrap0 = wrap

    def wrap(name, f):                      # noqa: F811  (the timers above plus the in-flight marker)
        g0 = _wrap0(name, f)

        def g(*a):
            inflight[0] = (name, time.perf_counter())
            try:
                return g0(*a)
            finally:
                inflight[0] = None
        return g
    for name in list(hip._fn):
        hip._fn[name] = wrap(name, hip._fn[name])
    sparse.read_ints = wrap('sparse.read_ints', sparse.read_ints)
    gcs = []
    t_gc = [0.0]

    def cb(phase, info):
        if phase == 'start':
            t_gc[0] = time.perf_counter()
        else:
            gcs.append((cur[0], info['generation'], round((time.perf_counter() - t_gc[0]) * 1e3, 2), info['collected']))
    if os.environ.get('HS_GC_LOG', '1') == '1':
        gc.callbacks.append(cb)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    wall = []
    ev[0].record()
    for i in range(steps):
        cur[0] = i
        h0 = time.perf_counter()
        last[0], last[1] = h0, 'step begin'
        step()
        ev[i + 1].record()
        wall.append((time.perf_counter() - h0) * 1e3)
    torch.cuda.synchronize()
    ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(steps)]
    med = sorted(ms)[steps // 2]
    print(f'{kind}: {steps} steps, median {med:.2f} ms, mean {sum(ms) / steps:.2f} ms; events over {thr} ms:')
    for i in range(steps):
        evs = [e for e in log if e[0] == i]
        g2 = [g for g in gcs if g[0] == i and (g[1] == 2 or g[2] >= 1.0)]
        flag = ' <== slow' if ms[i] > med * 1.15 else ''
        if flag or evs or g2:
            print(f'step {i:3d}: device {ms[i]:7.2f} ms  host {wall[i]:7.2f} ms{flag}')
            for e in evs:
                print(f'      {e[1]:14s} {e[2]:34s} {e[3]:8.2f} ms  {e[4]}')
            for g in g2:
                print(f'      gc generation {g[1]}: {g[2]} ms, {g[3]} collected')
    stop[0] = True
    if watch:
        print(f'\nwatchdog: {len(samples)} samples of calls running longer than {watch_ms} ms (main thread tid {main_tid})')
        prev = {}
        for st_, call_, el, snap, stack in samples:
            busy = []
            for tid, (comm, state, wchan, ticks) in sorted(snap.items()):
                d = ticks - prev.get(tid, ticks)
                prev[tid] = ticks
                if tid == main_tid or state == 'R' or d > 0:
                    busy.append(f'{"MAIN " if tid == main_tid else ""}{comm}[{tid}] {state} {wchan} +{d}')
            print(f'step {st_:3d} {call_:24s} +{el:6.1f} ms | ' + ' ; '.join(busy)[:700])
            if stack:
                print('        main kernel stack: ' + ' <- '.join(l.split()[-1] for l in stack.splitlines()[:8]))
    print('collections per generation:', [sum(1 for g in gcs if g[1] == k) for k in range(3)], ' gc.get_stats():', gc.get_stats())


if __name__ == '__main__':
    main()
