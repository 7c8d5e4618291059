"""Which GPU a batch or a pool of rnnoise_create() states goes to, against STUBBED device counts (VERDICT r5 Next #6b): no code
path with device >= 1 has ever run on hardware in this project -- the box has one GPU -- so the arithmetic at least is pinned here.
The functions are the ones the library calls (rnnoise_amd/csrc/device_choice.h, included by batch.cpp and dropin.cpp), compiled
for the host without HIP."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def prog(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("dc") / "device_choice_test")
    subprocess.run(["gcc", "-O1", "-Wall", "-Werror", os.path.join(ROOT, "tests", "csrc", "device_choice_test.c"), "-o", exe], check=True)
    def run(visible, pinned, n):
        out = subprocess.run([exe, str(visible), str(pinned), str(n)], capture_output=True, text=True, check=True).stdout
        pools, ok = out.split("|")
        return [int(x) for x in pools.split()], [int(x) for x in ok.split()]
    return run


def test_pools_rotate_over_the_visible_devices(prog):
    pools, _ = prog(8, -1, 20)
    assert pools == [k % 8 for k in range(20)]          # the k-th pool of a model on device k mod count
    pools, _ = prog(1, -1, 5)
    assert pools == [0] * 5
    pools, _ = prog(0, -1, 3)                            # no device: index 0, and the batch behind the pool fails loudly
    assert pools == [0] * 3


def test_a_pinned_device_takes_every_pool_and_is_clamped(prog):
    assert prog(8, 5, 4)[0] == [5] * 4
    assert prog(4, 7, 3)[0] == [3] * 3                   # $RNNOISE_AMD_DEVICE=7 on a 4-GPU node: the last one, not a failure
    assert prog(8, 0, 3)[0] == [0] * 3


@pytest.mark.parametrize("visible", [0, 1, 2, 8])
def test_batch_create_accepts_exactly_the_visible_indices(prog, visible):
    _, ok = prog(visible, -1, 0)
    assert ok == [0] + [1] * visible + [0]               # devices -1 .. visible: only 0 .. visible-1 pass


def test_the_library_uses_these_functions():
    src = {f: open(os.path.join(ROOT, "rnnoise_amd", "csrc", f)).read() for f in ("batch.cpp", "dropin.cpp")}
    assert "rn_device_index_ok(device, rnnoise_amd_device_count())" in src["batch.cpp"]
    assert "rn_pool_device(n_pools_so_far, pinned, rnnoise_amd_device_count())" in src["dropin.cpp"]
