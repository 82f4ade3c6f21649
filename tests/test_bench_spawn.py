"""`python bench.py --gpus N` without a launcher starts its own ranks: frave_amd.dist.spawn_ranks runs torch.distributed.run as a
child process, relays rank 0's line and refuses a line for another number of GPUs. CPU only: gloo and a stubbed step (tests/tools/stub_rank.py)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STUB = os.path.join(ROOT, "tests", "tools", "stub_rank.py")


def test_spawner_starts_n_ranks_and_relays_rank0s_line():
    from frave_amd.dist import spawn_ranks

    code, line = spawn_ranks(STUB, ["--gpus", "2", "--steps", "3"], 2, timeout=300)
    assert code == 0 and line["n_gpus"] == 2 and line["steps"] == 3
    assert line["elapsed"] >= 0.06 - 1e-3  # the MAX over ranks: rank 1 sleeps 2 x 0.01 x 3 s


def test_spawner_refuses_a_line_for_another_number_of_gpus():
    from frave_amd.dist import spawn_ranks

    code, line = spawn_ranks(STUB, ["--gpus", "2", "--lie", "1"], 2, timeout=300)
    assert code == 4 and line["n_gpus"] == 1


def test_spawner_relays_a_failing_rank():
    from frave_amd.dist import spawn_ranks

    code, line = spawn_ranks(STUB, ["--gpus", "2", "--fail"], 2, timeout=300)
    assert code != 0


def test_bench_py_without_launcher_spawns_ranks_and_fails_loudly_without_gpus():
    # No GPU here: both ranks die in frave_amd.Context / torch.cuda.set_device, the launcher exits non-zero and bench.py passes that on -
    # it must NOT print a one-GPU line (the round-4 behaviour) and must not hang.
    env = dict(os.environ, PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-extras"], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode != 0
    assert '"n_gpus"' not in r.stdout
    assert "spawn_ranks" in r.stderr


def test_bench_py_refuses_a_launcher_with_another_world_size():
    env = dict(os.environ, PYTHONPATH=ROOT, WORLD_SIZE="2", RANK="1", LOCAL_RANK="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--no-cpu-baseline"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                       timeout=600)
    assert r.returncode == 2 and "WORLD_SIZE=2" in r.stderr
