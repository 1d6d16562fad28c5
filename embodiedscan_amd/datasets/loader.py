"""Feeding the step from files: rank r of W reads its own shard of the scan list, a pool of host workers
decodes and draws the pipeline decisions, and hands over PINNED raw scans; the consumer copies them to HBM on its copy
stream (pipeline.upload_into) and everything else happens on the device.

Two worker kinds: `workers='thread'` (PIL releases the GIL inside the codecs, but numpy's legacy RandomState -- kept so
that the draws follow the reference's stream -- does not: the permutations of ~3e5 valid pixels per frame serialise, a few
scans/s at most) and `workers='process'` (forked workers that write the decoded arrays straight into shared-memory slots
the parent has registered as pinned: no pickling of pixels, no extra copy, scales with the cores).

Sharding follows mmengine's DefaultSampler, which the reference's dataloaders use (`sampler=dict(type='DefaultSampler',
shuffle=True)`, configs/detection/mv-det3d_...py:186; mmengine/dataset/sampler.py): a seeded permutation per epoch
(seed + epoch), padded by wrapping around to a multiple of the world size (round_up=True), rank r takes
indices[r::W] -- every rank gets the same number of scans and no data-path collective is needed.  `RepeatDataset(times=k)`
(same config, :188) is the `times` argument: indices are taken modulo the dataset length over a k-times longer range."""
import math
import queue
import threading

import numpy as np
import torch

from .. import pipeline


def effective_cpus(cgroup_root='/sys/fs/cgroup'):
    """host cores this process may actually use: the affinity mask capped by the cgroup CPU quota.  (The GPU boxes show 256
    logical CPUs and grant 16 cores -- `/sys/fs/cgroup/cpu.max` = `1600000 100000`: worker pools and CPU thread pools sized by
    os.cpu_count() oversubscribe that quota 8-16 x and run SLOWER, profiles/r4z_loader_*.json.)"""
    import os
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for quota_file, period_file in ((cgroup_root + '/cpu.max', None),
                                    (cgroup_root + '/cpu/cpu.cfs_quota_us', cgroup_root + '/cpu/cpu.cfs_period_us')):
        try:
            if period_file is None:
                q, per = open(quota_file).read().split()[:2]
            else:
                q, per = open(quota_file).read().strip(), open(period_file).read().strip()
            if q not in ('max', '-1') and int(per) > 0 and int(q) > 0:
                n = min(n, max(1, int(q) // int(per)))
            break
        except (OSError, ValueError):
            continue
    return max(1, n)


def shard_indices(n, rank=0, world=1, shuffle=True, seed=0, epoch=0, round_up=True, times=1):
    """this rank's scan indices for one epoch (DefaultSampler.__iter__ on a RepeatDataset of `times` repeats)"""
    total = n * times
    if shuffle:
        g = torch.Generator()
        g.manual_seed(seed + epoch)
        idx = torch.randperm(total, generator=g).tolist()
    else:
        idx = list(range(total))
    if round_up:
        num = math.ceil(total / world)
        size = num * world
        idx = (idx * int(size / len(idx) + 1))[:size] if total else []
    return [i % n for i in idx[rank::world]]


class _Batch(list):
    """a batch of pinned raw scans + the shared slots they live in (process mode), until ScanLoader.done()"""
    slots = None


class ScanLoader:
    """iterator over batches (lists of `batch_size` pinned raw scans) of this rank's shard.

    num_threads decode threads work ahead by `prefetch` scans; batches come out in shard order whatever the completion
    order of the threads, and every scan has its own RandomState seeded from (seed, epoch, position) so the decisions do
    not depend on thread scheduling."""

    def __init__(self, dataset, batch_size=4, rank=0, world=1, shuffle=True, seed=0, times=1, num_threads=8, prefetch=16,
                 pin=True, drop_last=True, workers='thread', slot_bytes=None, worker_timeout=120.0, exact_draws=None,
                 device_draws=None):
        """slot_bytes: size of one shared pinned slot in process mode (None: estimated from the frame headers of every
        source of the dataset -- EmbodiedScan mixes ScanNet / 3RScan / Matterport3D resolutions); a scan that still does
        not fit is decoded by the parent instead of aborting the epoch.  worker_timeout: seconds without any result
        after which the workers' liveness is checked (a worker killed by the OOM killer or a signal raises instead of
        hanging the training loop)."""
        assert workers in ('thread', 'process')
        if exact_draws is not None:          # None: whatever the dataset's pipeline says (default: the reference's exact stream);
            dataset.pipeline.exact_draws = bool(exact_draws)   # False: O(k) PointSample draws (loading.draw_without_order)
        if device_draws is not None:         # True: the host only decodes, both PointSample draws run on the GPU (pipeline.device_point_sample)
            dataset.pipeline.device_draws = bool(device_draws)
        self.dataset, self.batch_size = dataset, batch_size
        self.rank, self.world, self.shuffle, self.seed, self.times = rank, world, shuffle, seed, times
        self.num_threads, self.prefetch, self.pin, self.drop_last = max(1, num_threads), max(1, prefetch), pin, drop_last
        self.workers = workers
        self.slot_bytes, self.worker_timeout = slot_bytes, float(worker_timeout)
        self.epoch = 0
        self._slabs = None          # process mode: shared (and pinned) slots, allocated on first use, reused across epochs
        self._pending = []          # (event, slots) released by the consumer, reusable once the event has completed

    def set_epoch(self, epoch):
        self.epoch = epoch

    def indices(self):
        return shard_indices(len(self.dataset), self.rank, self.world, self.shuffle, self.seed, self.epoch, True, self.times)

    def __len__(self):
        n = len(self.indices())
        return n // self.batch_size if self.drop_last else -(-n // self.batch_size)

    def _rng(self, pos):
        return np.random.RandomState((self.seed * 1000003 + self.epoch * 7919 + pos * self.world + self.rank) % (2 ** 32))

    def _load(self, pos, idx):
        return pipeline.pin_scan(self.dataset.load_scan(idx, self._rng(pos)), pin=self.pin)

    def done(self, batch, event=None):
        """process mode: the consumer is finished with `batch`'s pinned buffers (event: a CUDA event recorded after the
        host->device copy was queued; the slots are reused once it has completed).  No-op in thread mode."""
        slots = getattr(batch, 'slots', None)
        if slots:
            batch.slots = None
            self._pending.append((event, slots))

    def __iter__(self):
        if self.workers == 'process':
            return self._iter_processes()
        return self._iter_threads()

    # ------------------------------------------------------------------ forked workers + shared pinned slots
    @staticmethod
    def _layout(host):
        """byte offsets (256-aligned) of the device-bound arrays of one scan inside a slot"""
        off, lay = 0, {}
        for k, v in host.items():
            lay[k] = (off, v.dtype, tuple(v.shape))
            off = (off + v.numel() * v.element_size() + 255) // 256 * 256
        return lay, off

    @staticmethod
    def _views(slab, lay):
        return {k: slab[o:o + int(np.prod(shape)) * torch.empty((), dtype=dt).element_size()].view(dt).view(shape)
                for k, (o, dt, shape) in lay.items()}

    def _estimate_slot_bytes(self, first_idx):
        """upper bound of a scan's device-bound bytes over the dataset: one scan is decoded (exact layout), and for every
        SOURCE of the dataset (first path component of sample_idx: scannet / 3rscan / matterport3d ...) the headers of one
        colour frame and one depth map give that source's frame sizes -- the slots are sized for the largest."""
        probe = pipeline._host_tensors(self.dataset.load_scan(first_idx, self._rng(0)))
        _, need = self._layout(probe)
        if 'sel_pix' not in probe:            # device-side draws: a scan that falls back to host draws also carries its index arrays
            need += 8 * int(self.dataset.pipeline.n_points) + 512
        V = int(probe['depth'].shape[0])
        fixed = need - int(probe['depth'].numel()) * 4 - int(probe['img_raw'].numel() if 'img_raw' in probe else probe['img'].numel())
        best = need
        try:
            from PIL import Image
            seen = set()
            for i in range(len(self.dataset)):
                info = self.dataset.get_data_info(i)
                src = str(info.get('sample_idx', '')).split('/')[0]
                if src in seen or not info.get('img_path'):
                    continue
                seen.add(src)
                with Image.open(info['img_path'][0]) as im:
                    w, h = im.size
                with Image.open(info['depth_img_path'][0]) as im:
                    wd, hd = im.size
                best = max(best, fixed + V * (h * w * 3 + hd * wd * 4) + 512 * V)
        except Exception:                                     # header probing is an optimisation: fall back to head-room
            best = max(best, int(need * 1.5))
        return int(best * 1.05) + 65536

    def _start_workers(self, first_idx):
        """allocate the shared slots, fork the PERSISTENT workers (the reference's `persistent_workers=True`), then pin
        the slots in the parent.  Order matters: the children must inherit the shared mappings, and forking is cheapest
        before the parent has registered gigabytes of pinned memory."""
        import multiprocessing as mp
        self._slot_bytes = int(self.slot_bytes) if self.slot_bytes else self._estimate_slot_bytes(first_idx)
        n_slots = self.prefetch + 2 * self.batch_size
        self._slabs = [torch.empty(self._slot_bytes, dtype=torch.uint8).share_memory_() for _ in range(n_slots)]
        ctx = mp.get_context('fork')
        self._tasks, self._results = ctx.Queue(), ctx.Queue()
        tasks, results, slabs = self._tasks, self._results, self._slabs
        ds, layout, views, slot_bytes = self.dataset, self._layout, self._views, self._slot_bytes

        def work():
            torch.set_num_threads(1)
            while True:
                t = tasks.get()
                if t is None:
                    return
                pos, i, slot, seed = t
                try:
                    slab = slabs[slot]

                    def alloc(V, ishape, dshape):
                        """frames are decoded INTO the slot: depth at offset 0, img_raw behind it -- the offsets `layout`
                        gives the first two arrays of pipeline._host_tensors (None: does not fit, the scan takes the
                        overflow path below)"""
                        nd = V * dshape[0] * dshape[1] * 4
                        o = (nd + 255) // 256 * 256
                        ni = V * ishape[0] * ishape[1] * 3
                        if o + ni + 65536 > slot_bytes:
                            return None
                        a = slab.numpy()
                        return (a[o:o + ni].reshape((V,) + tuple(ishape) + (3,)),
                                a[:nd].view(np.float32).reshape((V,) + tuple(dshape)))

                    scan = ds.load_scan(i, np.random.RandomState(seed), alloc)
                    host = pipeline._host_tensors(scan)
                    lay, need = layout(host)
                    if need > slot_bytes:                     # the parent decodes this one itself (slow path, no abort)
                        results.put((pos, slot, None, 'overflow', (i, seed, need)))
                        continue
                    dst = views(slab, lay)
                    for k, v in host.items():
                        if v.data_ptr() != dst[k].data_ptr():     # (the frames decoded in place are already there)
                            dst[k].copy_(v)
                    # small per-scan items travel as NUMPY arrays (pickled by value): a torch tensor on a multiprocessing queue
                    # goes through torch's shared-memory reduction -- a new shm segment and a file-descriptor hand-over per tensor,
                    # ~1 ms per scan of pure overhead in the consumer (round 5: 2.7 ms of loader wait per 4-scan step)
                    small = {k: (v.numpy() if torch.is_tensor(v) else v) for k, v in pipeline._finish({}, scan).items()}
                    results.put((pos, slot, lay, None, small))
                except Exception as e:                        # surfaced in the consumer
                    results.put((pos, slot, None, repr(e), None))

        self._procs = [ctx.Process(target=work, daemon=True) for _ in range(self.num_threads)]
        for p in self._procs:
            p.start()
        if self.pin and torch.cuda.is_available():
            for t in self._slabs:
                err = torch.cuda.cudart().cudaHostRegister(t.data_ptr(), t.numel(), 0)
                if int(err) != 0:
                    raise RuntimeError(f'cudaHostRegister failed with {int(err)}')
        self._free = list(range(n_slots))

    def close(self):
        """stop the persistent workers (process mode), wait for every copy that still reads a slot, forget the slot
        bookkeeping and un-register the pinned slabs before their shared mappings go away"""
        procs, self._procs = getattr(self, '_procs', None) or [], None
        for _ in procs:
            self._tasks.put(None)
        for p in procs:
            p.join(timeout=5)
            if p.is_alive():
                p.terminate()
        for ev, _ in self._pending:                           # copies out of the slots must have finished
            if ev is not None:
                try:
                    ev.synchronize()
                except Exception:
                    pass
        self._pending = []                                    # stale slot ids must not leak into a later epoch's free list
        self._free = []
        slabs, self._slabs = self._slabs, None
        if slabs and self.pin and torch.cuda.is_available():
            try:
                torch.cuda.synchronize()
                rt = torch.cuda.cudart()
                for t in slabs:                               # the registration must not outlive the mapping
                    rt.cudaHostUnregister(t.data_ptr())
            except Exception:
                pass

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _get_result(self):
        """next worker result; every `worker_timeout` seconds of silence the workers' liveness is checked"""
        while True:
            try:
                return self._results.get(timeout=self.worker_timeout)
            except queue.Empty:
                dead = [p.pid for p in (self._procs or []) if not p.is_alive()]
                if dead or not self._procs:
                    raise RuntimeError(f'ScanLoader: worker process(es) {dead} died (killed by the OOM killer or a signal?); '
                                       'no result for %.0f s' % self.worker_timeout)

    def _reclaim(self):
        keep = []
        for ev, sl in self._pending:
            if ev is None or ev.query():
                self._free.extend(sl)
            else:
                keep.append((ev, sl))
        self._pending = keep

    def _iter_processes(self):
        idx = self.indices()
        n_batches = len(self)
        idx = idx[:n_batches * self.batch_size] if self.drop_last else idx
        if not idx:
            return
        if self._slabs is None:
            self._start_workers(idx[0])
        slabs, free = self._slabs, self._free
        nxt, got, in_flight = 0, {}, 0
        try:
            for b in range(n_batches):
                lo, hi = b * self.batch_size, min(len(idx), (b + 1) * self.batch_size)
                while any(pos not in got for pos in range(lo, hi)):
                    self._reclaim()
                    while nxt < len(idx) and free and in_flight < self.prefetch:
                        seed = (self.seed * 1000003 + self.epoch * 7919 + nxt * self.world + self.rank) % (2 ** 32)
                        self._tasks.put((nxt, idx[nxt], free.pop(), seed))
                        nxt, in_flight = nxt + 1, in_flight + 1
                    if in_flight == 0:
                        if self._pending:                     # every slot waits for a copy to finish
                            self._pending[0][0].synchronize()
                            continue
                        raise RuntimeError('ScanLoader: every pinned slot is held by a batch the consumer has not '
                                           'released -- call loader.done(batch[, event]) after queueing its copy')
                    pos, slot, lay, err, small = self._get_result()
                    in_flight -= 1
                    if err == 'overflow':                     # larger than a slot: decode here into its own pinned buffers
                        free.append(slot)
                        i, seed, need = small
                        got[pos] = (None, None, pipeline.pin_scan(self.dataset.load_scan(i, np.random.RandomState(seed)),
                                                                  pin=self.pin))
                        continue
                    if err is not None:
                        free.append(slot)
                        raise RuntimeError(f'scan {idx[pos]} (position {pos}): {err}')
                    got[pos] = (slot, lay, small)
                batch, slots = _Batch(), []
                for pos in range(lo, hi):
                    slot, lay, small = got.pop(pos)
                    if slot is None:
                        batch.append(small)
                        continue
                    d = self._views(slabs[slot], lay)
                    d.update({k: (torch.from_numpy(v) if isinstance(v, np.ndarray) and k in ('gt_boxes', 'gt_labels') else v)
                              for k, v in small.items()})
                    batch.append(d)
                    slots.append(slot)
                batch.slots = slots
                yield batch
        finally:
            # an abandoned epoch: drain what is still in flight so that the slots come back
            for slot, _, _ in got.values():
                if slot is not None:
                    free.append(slot)
            while in_flight > 0:
                try:
                    _, slot, _, _, _ = self._results.get(timeout=30)
                except Exception:
                    break
                free.append(slot)
                in_flight -= 1

    # ------------------------------------------------------------------ threads
    def _iter_threads(self):
        idx = self.indices()
        n_batches = len(self)
        idx = idx[:n_batches * self.batch_size] if self.drop_last else idx
        done, lock, cv = {}, threading.Lock(), threading.Condition()
        todo = queue.Queue()
        for pos, i in enumerate(idx):
            todo.put((pos, i))
        state = dict(next=0, stop=False, err=None)

        def worker():
            while not state['stop']:
                try:
                    pos, i = todo.get_nowait()
                except queue.Empty:
                    return
                with cv:                                  # stay at most `prefetch` scans ahead of the consumer
                    while pos >= state['next'] + self.prefetch and not state['stop']:
                        cv.wait(0.05)
                try:
                    scan = self._load(pos, i)
                except Exception as e:                    # surfaced in the consumer thread
                    scan, state['err'] = None, e
                with cv:
                    done[pos] = scan
                    cv.notify_all()

        threads = [threading.Thread(target=worker, daemon=True) for _ in range(self.num_threads)]
        for t in threads:
            t.start()
        try:
            for b in range(n_batches):
                batch = []
                for pos in range(b * self.batch_size, min(len(idx), (b + 1) * self.batch_size)):
                    with cv:
                        while pos not in done and state['err'] is None:
                            cv.wait(0.05)
                        if state['err'] is not None:
                            raise state['err']
                        batch.append(done.pop(pos))
                        state['next'] = pos + 1
                        cv.notify_all()
                yield batch
        finally:
            state['stop'] = True
            with cv:
                cv.notify_all()
            for t in threads:
                t.join()
