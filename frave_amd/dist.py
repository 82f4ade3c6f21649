"""Multi-GPU control flow of the benchmark / batch drivers: which images a rank owns, and the timing protocol.

The path shards by image with no data-path collective: image i belongs to rank i mod world.
torch.distributed is used for the barrier and the max-over-ranks of the elapsed time only; `backend` is "nccl" (= RCCL)
on GPUs and "gloo" in the CPU tests.
"""
import time


def images_for_rank(n_images, rank, world):
    """Indices of the images rank `rank` of `world` processes owns (BASELINE config 4): the library's partition
    (fri_hip_shard_size / fri_hip_shard_image, include/fri_hip.h: image i -> shard i mod world), the same one the
    one-process multi-GPU helper fri_hip_multi_transform_quant and `fri_driver batch --gpus N` use."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError("bad rank/world")
    from .api import shard_images

    return shard_images(n_images, rank, world)


def timed_region(fn, dist=None, device_sync=None, device=None):
    """Runs fn() bracketed by barrier + device synchronisation on both sides; returns the MAX elapsed seconds over ranks."""
    import torch

    def fence():  # this rank's device work is finished, then every rank's is (a barrier on the host; without it one synchronise is all there is to do)
        if device_sync:
            device_sync()
        if dist is not None:
            dist.barrier()
            if device_sync:
                device_sync()  # the barrier of the nccl backend is itself device work

    fence()
    t0 = time.perf_counter()
    fn()
    fence()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device or "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed
