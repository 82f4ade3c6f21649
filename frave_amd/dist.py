"""Multi-GPU control flow of the benchmark / batch drivers: which images a rank owns, and the timing protocol.

The path shards by image with no data-path collective: image i belongs to rank i mod world.
torch.distributed is used for the barrier and the max-over-ranks of the elapsed time only; `backend` is "nccl" (= RCCL)
on GPUs and "gloo" in the CPU tests.
"""
import json
import os
import socket
import subprocess
import sys
import time


def images_for_rank(n_images, rank, world):
    """Indices of the images rank `rank` of `world` processes owns (BASELINE config 4): the library's partition
    (fri_hip_shard_size / fri_hip_shard_image, include/fri_hip.h: image i -> shard i mod world), the same one the
    one-process multi-GPU helper fri_hip_multi_transform_quant and `fri_driver batch --gpus N` use."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError("bad rank/world")
    from .api import shard_images

    return shard_images(n_images, rank, world)


def timed_region(fn, dist=None, device_sync=None, device=None, device_idle=None):
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
    # closing fence: the rank's device work is finished. `device_idle` (optional) is a non-blocking proof of that - hipStreamQuery on the launch
    # stream; fn() of the benchmark returns only after it has polled its end event - and spares the full device synchronise, which costs ~160 us
    # on an already idle device whenever a profiler is attached. Not idle (or no probe): synchronise as before.
    if not (device_idle and device_idle()):
        sync()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        dist.barrier()
        sync()
        t = torch.tensor([elapsed], dtype=torch.float64, device=device or "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(script, argv, n_ranks, env=None, timeout=None):
    """`python script --gpus N ...` started WITHOUT a launcher: run `python -m torch.distributed.run --nnodes=1 --nproc-per-node N
    --master-addr 127.0.0.1 --master-port <free> script argv...` as a CHILD process (never an exec: the caller may not replace itself, and
    must not have touched the GPU before calling this), relay the ranks' stderr, find rank 0's ONE JSON line on the child's stdout and check
    that it reports `n_gpus == n_ranks`. Returns (exit_code, line_dict_or_None): non-zero when the launcher failed, when no JSON line
    came back, or when the line is for another number of GPUs - a one-GPU number must never pass for an N-GPU one.
    The per-image loop this shards: crates/fri-cli/src/commands/bench.rs:15-120 (image i -> rank i mod N, no collective)."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_ranks}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), script] + list(argv)
    child_env = dict(os.environ if env is None else env)
    child_env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs on this driver
    child_env.setdefault("OMP_NUM_THREADS", "1")
    proc = subprocess.run(cmd, env=child_env, stdout=subprocess.PIPE, text=True, timeout=timeout)
    line = None
    for ln in proc.stdout.splitlines():
        ln = ln.strip()
        if ln.startswith("{") and ln.endswith("}"):
            try:
                cand = json.loads(ln)
            except ValueError:
                continue
            if "n_gpus" in cand:
                line = cand
    if proc.returncode != 0:
        print(f"spawn_ranks: the launcher exited with {proc.returncode}", file=sys.stderr)
        return proc.returncode, line
    if line is None:
        print("spawn_ranks: no JSON line from rank 0", file=sys.stderr)
        return 3, None
    if line.get("n_gpus") != n_ranks:
        print(f"spawn_ranks: asked for {n_ranks} ranks but the line says n_gpus={line.get('n_gpus')}", file=sys.stderr)
        return 4, line
    return 0, line
