import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_sessionstart(session):
    """Keep every native piece in step with its sources (no-ops when up to date; hipcc cross-compiles without a GPU)."""
    import subprocess
    for d in ("raytracer_challenge_amd/csrc", "oracle", "tests/cpu_emu"):
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, d)], check=True)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    from oracle_lib import oracle
    return oracle()


@pytest.fixture(scope="session")
def hip():
    import raytracer_challenge_amd as rt
    return rt.hip_backend()
