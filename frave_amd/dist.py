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
    """Runs fn() bracketed by barrier + device synchronisation on both sides; returns the MAX over ranks of the seconds a rank needed
    from leaving the opening fence to having finished its own device work. The closing barrier is there (nobody leaves before
    everybody is done) but its own latency - 150-250 us for an RCCL barrier, as much as twenty 17-us steps - is not charged to the
    steps: the clock of a rank stops when its synchronise returns, and the maximum over ranks is the job's time."""
    import torch

    def sync():
        if device_sync:
            device_sync()

    sync()  # opening fence: this rank is idle, then every rank is (the barrier of the nccl backend is itself device work)
    if dist is not None:
        dist.barrier()
        sync()
    t0 = time.perf_counter()
    fn()
    sync()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        dist.barrier()
        sync()
        t = torch.tensor([elapsed], dtype=torch.float64, device=device or "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed
