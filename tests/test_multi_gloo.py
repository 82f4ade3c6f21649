"""N > 1 control flow on CPU: world_size 2 over gloo. Covers the image -> rank partition (no data-path collective) and the
barrier / max-over-ranks timing protocol bench.py uses."""
import os
import socket

import pytest


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_images, out):
    import torch
    import torch.distributed as dist

    from frave_amd.dist import images_for_rank, timed_region

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = images_for_rank(n_images, rank, world)
    # every rank builds the same host-only plan (plans are replicated, never communicated)
    import frave_amd

    plan = frave_amd.Plan(None, 320, 200, 1)
    work = {"n": 0}

    def fn():
        import time

        for _ in mine:
            work["n"] += plan.num_cells
        time.sleep(0.05 * (rank + 1))  # rank 1 is slower: the reported time must be the max

    elapsed = timed_region(fn, dist=dist)
    gathered = [None] * world
    dist.all_gather_object(gathered, (mine, work["n"], elapsed))
    if rank == 0:
        out.put(gathered)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_images", [7, 1024])
def test_two_ranks_partition_and_timing(n_images):
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_images, out)) for r in range(2)]
    for p in procs:
        p.start()
    gathered = out.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    owned = sorted(i for mine, _, _ in gathered for i in mine)
    assert owned == list(range(n_images))  # disjoint and complete
    assert abs(len(gathered[0][0]) - len(gathered[1][0])) <= 1
    t0, t1 = gathered[0][2], gathered[1][2]
    assert t0 == t1 and t0 >= 0.1  # both ranks report the max (the slower rank slept 0.1 s)


def test_partition_edge_cases():
    from frave_amd.dist import images_for_rank

    assert images_for_rank(0, 0, 1) == []
    assert images_for_rank(3, 2, 8) == [2] and images_for_rank(3, 5, 8) == []
    assert [len(images_for_rank(1024, r, 8)) for r in range(8)] == [128] * 8
    with pytest.raises(ValueError):
        images_for_rank(4, 2, 2)
