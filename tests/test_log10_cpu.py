"""The feature stage's log10 (src/denoise.c:383) is the host libm's: rnnoise_amd/csrc/log10_glibc.h restates GNU libc's
__ieee754_log10 / __log (FMA build) operation for operation.  Here, without a GPU: the header compiled for the HOST (the same
source the device compiles; every operation in it is an IEEE double add / multiply / fma) against the running libm --
EXHAUSTIVELY over every non-negative float band energy, and on random doubles for the functions themselves; the constants
against the live libm; the library's model selection."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "csrc", "log10_sweep.c")


@pytest.fixture(scope="module")
def sweep(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("log10") / "log10_sweep")
    subprocess.run(["gcc", "-O2", "-mfma", "-ffp-contract=off", "-o", exe, SRC, "-lm"], check=True)

    def run(*args):
        n, bad, badf, first = subprocess.run([exe, *map(str, args)], check=True, capture_output=True, text=True).stdout.split()
        return int(n), int(bad), int(badf), first
    return run


def _glibc_is_the_modelled_one():
    import platform
    lib, ver = platform.libc_ver()
    return lib == "glibc" and tuple(map(int, ver.split(".")[:2])) >= (2, 28) and "fma" in open("/proc/cpuinfo").read()


needs_glibc = pytest.mark.skipif(not _glibc_is_the_modelled_one(), reason="host libm is not GNU libc >= 2.28 on an FMA machine")


@needs_glibc
def test_every_float_band_energy(sweep):
    """x = 1e-2 + (double)Ex for EVERY float Ex in [0, +Inf]: the restatement returns the libm's double, bit for bit"""
    edges = [i << 27 for i in range(16)] + [0x7f800001]
    with ThreadPoolExecutor(os.cpu_count() or 4) as ex:
        res = list(ex.map(lambda i: sweep(5, edges[i], edges[i + 1]), range(16)))
    assert sum(r[0] for r in res) == 0x7f800001
    assert all(r[1] == 0 for r in res), [r for r in res if r[1]]


@needs_glibc
@pytest.mark.parametrize("mode,what", [(0, "band energies, random mantissas"), (1, "log10, random normal doubles"),
                                        (2, "log10 around 1 (both sides of the near-1 branch)"), (3, "log() itself"),
                                        (4, "zeros, subnormals, negatives, Inf, NaN, branch edges")])
def test_restatement_equals_the_libm(sweep, mode, what):
    n, bad, badf, first = sweep(mode, 4_000_000, 11)
    assert bad == 0, f"{what}: {bad} of {n} doubles differ, first argument {first}"


@needs_glibc
def test_constants_are_the_live_libm_s():
    """log10_glibc_data.h is what tools/extract_glibc_log_table.py reads out of this machine's libm.so.6"""
    import ctypes.util
    path = None
    for line in open("/proc/self/maps"):
        if "libm.so" in line or "libm-" in line:
            path = line.split()[-1]
            break
    if path is None:
        import ctypes
        ctypes.CDLL(ctypes.util.find_library("m"))
        path = next(line.split()[-1] for line in open("/proc/self/maps") if "libm.so" in line or "libm-" in line)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "extract_glibc_log_table.py"), path], check=True,
                         capture_output=True, text=True).stdout
    have = open(os.path.join(ROOT, "rnnoise_amd", "csrc", "log10_glibc_data.h")).read()
    strip = lambda s: [ln for ln in s.splitlines() if not ln.startswith("//")]
    assert strip(out) == strip(have)


def test_model_selection():
    from rnnoise_amd import capi
    name = capi.log10_model()
    assert name in ("host=glibc-fma", "host=unknown:ocml", "glibc-fma", "ocml")
    if _glibc_is_the_modelled_one() and not os.environ.get("RNNOISE_AMD_LOG10"):
        assert name == "host=glibc-fma"
    code = "from rnnoise_amd import capi; print(capi.log10_model())"
    for want in ("ocml", "glibc-fma"):
        got = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, RNNOISE_AMD_LOG10=want, PYTHONPATH=ROOT), check=True,
                             capture_output=True, text=True).stdout.strip()
        assert got == want
