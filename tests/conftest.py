import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """A fresh checkout has no built artefacts (they are git-ignored): build the product library before the first test needs it.
    hipcc cross-compiles for gfx950 without a GPU; this is `__graft_entry__.build()` minus the import check."""
    import subprocess

    if not os.path.exists(os.path.join(ROOT, "frave_amd", "libfri_hip.so")):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "frave_amd", "csrc")])


@pytest.fixture(scope="session")
def oracle():
    from oracle import fri_oracle

    fri_oracle.build()
    return fri_oracle
