"""Shared fixtures.  `-m "not gpu"`: the oracles against their pins, the host
logic and the C-ABI surface (no GPU, a few minutes at most).  `-m gpu`: the
parity tests proper -- every one calls librtiow_hip.so through its C ABI and
compares with the CPU oracle (tests are the only place that loads oracle/)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_mod():
    import oracle
    oracle.load()
    return oracle


@pytest.fixture(scope="session")
def book1_flat():
    """The committed flat scene (tests/golden/book1_scene_seed1.npy)."""
    import rtiow_amd as rt
    a = np.load(os.path.join(GOLDEN, "book1_scene_seed1.npy"), allow_pickle=False)
    return np.ascontiguousarray(a.astype(rt.SPHERE_DTYPE))


@pytest.fixture(scope="session")
def golden_small():
    return np.load(os.path.join(GOLDEN, "book1_32x18_4spp.npz"), allow_pickle=False)


@pytest.fixture(scope="session")
def renderer():
    """One rt_context on cuda:0 for the GPU tests (fails loudly without a gfx950)."""
    import rtiow_amd as rt
    r = rt.Renderer(0)
    yield r
    r.close()
