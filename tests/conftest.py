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


def splitmix64_bytes(seed, n):
    """SURVEY.md §8d input generator (ii): splitmix64(seed), 8 little-endian bytes per draw."""
    cnt = (n + 7) // 8
    state = (np.uint64(seed) + np.uint64(0x9E3779B97F4A7C15) * np.arange(1, cnt + 1, dtype=np.uint64)).astype(np.uint64)
    z = state
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    z = z ^ (z >> np.uint64(31))
    return z.view(np.uint8)[:n].copy()


def pattern_bytes(n):
    """The reference bench pattern (benches/commit.rs:6-8): (i % 256) as u8."""
    return (np.arange(n, dtype=np.uint64) % 256).astype(np.uint8)


@pytest.fixture(scope="session")
def blob():
    with open(os.path.join(GOLDEN, "blob"), "rb") as f:
        return f.read()


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O

    O.build()
    return O


@pytest.fixture(scope="session")
def gpu_ctx():
    import torch

    assert torch.cuda.is_available(), "-m gpu tests need a GPU"
    import frieda_amd

    ctx = frieda_amd.Context(0)
    yield ctx
    ctx.close()
