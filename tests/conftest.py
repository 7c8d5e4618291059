import lzma
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu through gpurun)")


def load_blob(name="default") -> bytes:
    with open(os.path.join(GOLD, f"{name}.blob.xz"), "rb") as f:
        return lzma.decompress(f.read())


@pytest.fixture(scope="session")
def blob_default():
    return load_blob("default")


@pytest.fixture(scope="session")
def blob_little():
    return load_blob("little")


def golden(name):
    return np.load(os.path.join(GOLD, name))


def bits(a):
    """float arrays compared as bit patterns (NaN-safe, -0 != +0)."""
    a = np.ascontiguousarray(a)
    return a.view(np.uint32) if a.dtype == np.float32 else a


def assert_bits_equal(a, b, what=""):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, f"{what}: shape {a.shape} vs {b.shape}"
    ne = bits(a) != bits(b)
    if ne.any():
        idx = np.argwhere(ne)[:5].tolist()
        raise AssertionError(f"{what}: {int(ne.sum())} of {ne.size} words differ, first at {idx}: "
                             f"{a[tuple(np.argwhere(ne)[0])]!r} vs {b[tuple(np.argwhere(ne)[0])]!r}")
