"""Feeding the step from files: rank r of W reads its own shard of the scan list, a pool of host threads
decodes (PIL releases the GIL inside the codecs) and draws the pipeline decisions, and hands over PINNED raw scans; the
consumer copies them to HBM on its copy stream (pipeline.upload_into) and everything else happens on the device.

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


class ScanLoader:
    """iterator over batches (lists of `batch_size` pinned raw scans) of this rank's shard.

    num_threads decode threads work ahead by `prefetch` scans; batches come out in shard order whatever the completion
    order of the threads, and every scan has its own RandomState seeded from (seed, epoch, position) so the decisions do
    not depend on thread scheduling."""

    def __init__(self, dataset, batch_size=4, rank=0, world=1, shuffle=True, seed=0, times=1, num_threads=8, prefetch=16,
                 pin=True, drop_last=True):
        self.dataset, self.batch_size = dataset, batch_size
        self.rank, self.world, self.shuffle, self.seed, self.times = rank, world, shuffle, seed, times
        self.num_threads, self.prefetch, self.pin, self.drop_last = max(1, num_threads), max(1, prefetch), pin, drop_last
        self.epoch = 0

    def set_epoch(self, epoch):
        self.epoch = epoch

    def indices(self):
        return shard_indices(len(self.dataset), self.rank, self.world, self.shuffle, self.seed, self.epoch, True, self.times)

    def __len__(self):
        n = len(self.indices())
        return n // self.batch_size if self.drop_last else -(-n // self.batch_size)

    def _load(self, pos, idx):
        rng = np.random.RandomState((self.seed * 1000003 + self.epoch * 7919 + pos * self.world + self.rank) % (2 ** 32))
        return pipeline.pin_scan(self.dataset.load_scan(idx, rng), pin=self.pin)

    def __iter__(self):
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
