"""Data-parallel exchange steps of the train step: scans are independent units, every rank
processes its own scans end to end; the only collectives are ONE all-reduce of the flat gradient arena per step and
ONE all-reduce of the per-sample positive counts (the reference issues `reduce_mean` once per sample inside a Python
loop, embodiedscan/utils/dist_utils.py:4-10 called from dense_heads/fcaf3d_head.py:1183).
Backend: "nccl" (== RCCL over xGMI on ROCm) on GPUs, "gloo" in the CPU tests."""
import torch
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
