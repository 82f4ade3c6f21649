"""A stand-in for bench.py's rank body used by tests/test_bench_spawn.py: the same launcher environment (RANK / WORLD_SIZE / MASTER_*), the same
timing protocol (frave_amd.dist.timed_region: barrier on both sides, MAX over ranks) over gloo, a stubbed step (a sleep) instead of a kernel.
`--lie N` makes rank 0 print n_gpus = N (the spawner must refuse a line for another number of GPUs); `--fail` makes rank 1 exit non-zero."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=0)
    ap.add_argument("--lie", type=int, default=0)
    ap.add_argument("--fail", action="store_true")
    a = ap.parse_args()
    world, rank = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"])
    if world != a.gpus:
        sys.exit(2)
    import torch.distributed as dist

    from frave_amd.dist import images_for_rank, timed_region

    dist.init_process_group("gloo")
    if a.fail and rank == 1:
        sys.exit(7)
    mine = images_for_rank(world * a.steps, rank, world)
    elapsed = timed_region(lambda: time.sleep(0.01 * len(mine) * (rank + 1)), dist=dist)
    if rank == 0:
        print("noise on stdout that is not the line")
        print(json.dumps({"metric": "stub", "value": world * a.steps / elapsed, "n_gpus": a.lie or world, "steps": a.steps, "elapsed": elapsed}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
