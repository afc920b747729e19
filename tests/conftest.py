import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
# Every test process runs the engine with its device-affinity assertion on (csrc/context.h device_affinity_ok): a HIP call made for a
# context while another device is current fails the call instead of passing silently on a one-GPU box.
os.environ.setdefault("BBP_DEBUG_DEVICE_CHECK", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """Make sure every native artefact exists (idempotent; no-op when the prebuilt files travelled with the repo)."""
    import __graft_entry__ as ge
    ge.build_hip()
    ge.build_oracle()
    ge.build_hostcheck()
    return ge


@pytest.fixture(scope="session")
def bbp(built):
    # torch ships its own copy of the HIP runtime; when libbbp_hip.so (linked against /opt/rocm) is loaded BEFORE torch and
    # torch then initialises the GPU, the second runtime reports zero devices.  Load torch's first, whatever subset of tests runs.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    import dusk_blindbidproof_amd as m
    return m


@pytest.fixture(scope="session")
def ctx(bbp):
    try:  # tests that also use torch streams need torch's HIP runtime initialised before ours is (bench.py's order)
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except ImportError:
        pass
    c = bbp.Context(0)  # raises loudly without a gfx950 device: no fallback
    yield c
    flags = c.health()  # after the whole GPU session: no MSM gather was ever out of range (engine scratch never corrupted)
    c.close()
    assert flags == 0, "engine health flags %#x" % flags


@pytest.fixture(scope="session")
def golden():
    import json

    def load(name):
        with open(os.path.join(GOLDEN, name)) as f:
            return json.load(f)
    return load
