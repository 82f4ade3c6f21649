"""Randomised parity on the GPU: shapes (incl. thin and tiny images), contents, channel counts, quantisers and predictor
parameters drawn from a fixed seed, every kernel against the CPU oracle (tests/tools/fuzz_parity.py runs longer sweeps)."""
import pytest


@pytest.mark.gpu
def test_random_cases_match_the_oracle():
    from tests.tools.fuzz_parity import run

    assert run(80, 20261004) == 0
