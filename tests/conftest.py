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
    config.addinivalue_line("markers", "rcp(name): rcpps profile of the oracle and the product for this test (default intel)")


def use_rcp_profile(name: str) -> None:
    """one `rcpps` profile for the oracle and -- where the HIP library is built -- the product (include/rnnoise_amd.h)"""
    from oracle import binding
    binding.set_rcp_profile(name)
    from rnnoise_amd import capi
    if os.path.exists(capi.LIB_PATH):
        capi.set_rcp_profile(name)
    os.environ["RNNOISE_AMD_RCP_PROFILE"] = name  # child processes that load the library (demo binaries, CLI, rank workers)


@pytest.fixture(autouse=True)
def rcp_profile(request):
    """Every test runs on the profile of the committed goldens ("intel", the build host's CPU family) unless it asks for
    another one with @pytest.mark.rcp("host" | "amd-zen5"): live comparisons against the compiled reference need "host"."""
    m = request.node.get_closest_marker("rcp")
    name = getattr(request, "param", None) or (m.args[0] if m else "intel")
    use_rcp_profile(name)
    return name


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
    """a committed reference-output fixture; it belongs to the rcpps profile of the host that produced it, which must be
    the profile the test runs on (tests/golden/make_golden.py: host_profile)"""
    from oracle import binding
    prof = binding.rcp_profile()
    path = os.path.join(GOLD, name) if prof == "intel" else os.path.join(GOLD, prof.replace("-", "_"), name)
    if not os.path.exists(path):
        pytest.skip(f"no {name} fixture for the {prof} profile")
    g = np.load(path)
    if "rcp_profile" in g.files:
        assert str(g["rcp_profile"]) == prof, f"{name} was made on the {g['rcp_profile']} profile"
    return g


# golden tests run once per CPU family that has committed reference outputs (tests/golden/, tests/golden/amd_zen5/)
both_profiles = pytest.mark.parametrize("rcp_profile", ["intel", "amd-zen5"], indirect=True)


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
