"""Data-parallel exchange steps of the train step: scans are independent units, every rank
processes its own scans end to end; the only collectives are ONE all-reduce of the flat gradient arena per step and
ONE all-reduce of the per-sample positive counts (the reference issues `reduce_mean` once per sample inside a Python
loop, embodiedscan/utils/dist_utils.py:4-10 called from dense_heads/fcaf3d_head.py:1183).
Backend: "nccl" (== RCCL over xGMI on ROCm) on GPUs, "gloo" in the CPU tests."""
import torch.distributed as dist


def is_dist():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def allreduce_mean_(flat):
    """in-place mean over ranks of a flat tensor (the gradient arena)."""
    if is_dist():
        dist.all_reduce(flat)
        flat.mul_(1.0 / dist.get_world_size())
    return flat


def reduce_mean(t):
    """embodiedscan.utils.dist_utils.reduce_mean for a whole vector at once (not in place)."""
    if not is_dist():
        return t
    t = t.clone() / dist.get_world_size()
    dist.all_reduce(t)
    return t


class BucketedGradReducer:
    """Overlaps the gradient all-reduce with the backward pass.

    The gradient arena is laid out [2-D backbone | 3-D backbone | everything else]; backward finishes those parts in the opposite
    order, so each part is all-reduced asynchronously (torch.distributed `async_op=True`: RCCL runs it on its own
    stream after an event on the compute stream) as soon as the tape has passed the marker behind it: the 86 MB head
    bucket travels over xGMI under the 3-D backbone's backward, the 254 MB 3-D bucket under the 2-D backbone's.
    The reference gets the same effect from DDP's bucketed reducer (mmengine MMDistributedDataParallel)."""

    def __init__(self, arena, prefixes=('backbone.', 'backbone_3d.')):
        """parts 0..len(prefixes)-1 = the trainable tensors under each prefix; the LAST part = everything else (heads,
        necks, decoders: whatever the detector defines after its backbones).  The arena packs trainable tensors in spec
        order, so every part is one contiguous range and the parts tile [0, n_train)."""
        self.arena = arena
        names = arena.trainable_names()
        spans = []
        for pre in prefixes:
            offs = [arena.offsets[n] for n in names if n.startswith(pre)]
            spans.append((min(o for o, _ in offs), max(o + ((n + 3) // 4) * 4 for o, n in offs)) if offs else None)
        rest = [arena.offsets[n] for n in names if not any(n.startswith(p) for p in prefixes)]
        spans.append((min(o for o, _ in rest), max(o + ((n + 3) // 4) * 4 for o, n in rest)) if rest else None)
        self.ranges = [sp if sp is not None else (0, 0) for sp in spans]
        real = sorted(sp for sp in spans if sp is not None)
        assert real and real[0][0] == 0 and real[-1][1] == arena.n_train and \
            all(a[1] == b[0] for a, b in zip(real[:-1], real[1:])), f'gradient buckets do not tile the arena: {real}'
        self.work = []

    def launch(self, part):
        """all-reduce (sum) part `part` of the gradient arena without blocking the compute stream"""
        a, b = self.ranges[part]
        if is_dist() and b > a:
            self.work.append(dist.all_reduce(self.arena.grad[a:b], async_op=True))

    def finish(self):
        """wait for every launched bucket and turn the sums into means; returns True if anything was reduced"""
        if not self.work:
            return False
        for w in self.work:
            w.wait()
        self.work = []
        self.arena.grad.mul_(1.0 / dist.get_world_size())
        return True
