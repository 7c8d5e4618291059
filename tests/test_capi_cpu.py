"""CPU suite, part 2: the C-ABI boundary without a GPU.

librnnoise_amd.so must load, export every symbol that include/rnnoise.h and
include/rnnoise_amd.h declare (the reference CI's `nm` check, .gitlab-ci.yml:10-41), run its
host-only logic (blob parsing, weight-byte accounting) and FAIL LOUDLY -- not fall back to
a CPU path -- when no HIP device is visible.  No compute call is made here.
"""
import ctypes as C
import os
import re
import struct
import subprocess

import numpy as np
import pytest

from conftest import ROOT
from rnnoise_amd import capi

HAVE_GPU = capi.lib().rnnoise_amd_device_count() > 0 if os.path.exists(capi.LIB_PATH) else False


def declared_symbols():
    names = []
    for h in ("rnnoise.h", "rnnoise_amd.h"):
        src = open(os.path.join(ROOT, "include", h)).read()
        names += re.findall(r"RNNOISE_EXPORT\s+[\w\s\*]+?\b(rnnoise_\w+)\s*\(", src)
    return names


def test_library_exports_every_declared_symbol():
    L = capi.lib()
    decl = declared_symbols()
    assert len(decl) >= 28 and sorted(decl) == sorted(capi.EXPORTS)
    for n in decl:
        assert hasattr(L, n), n
    nm = subprocess.run(["nm", "-D", "--defined-only", capi.LIB_PATH], capture_output=True, text=True).stdout
    for n in decl:
        assert re.search(rf"\bT {n}\b", nm), f"{n} not a defined text symbol"


def test_product_library_exports_the_api_and_nothing_else():
    """librnnoise_amd.so and librnnoise.so.0: every dynamic symbol is one declared in include/rnnoise.h / rnnoise_amd.h -- no
    launch helpers, kernel stubs, probe kernels or debug taps (those live in the instrumented library, whose extra entry
    points are the ones of include/rnnoise_amd_debug.h)"""
    decl = set(declared_symbols())
    for so in (capi.LIB_PATH, os.path.join(ROOT, "rnnoise_amd", "librnnoise.so.0")):
        nm = subprocess.run(["nm", "-D", "--defined-only", so], capture_output=True, text=True).stdout
        syms = {l.split()[-1] for l in nm.splitlines() if l.strip()}
        assert syms == decl, (so, sorted(syms ^ decl))
    dbg = re.findall(r"RNNOISE_EXPORT\s+[\w\s\*]+?\b(rnnoise_\w+)\s*\(", open(os.path.join(ROOT, "include", "rnnoise_amd_debug.h")).read())
    assert sorted(dbg) == sorted(capi.DEBUG_EXPORTS)
    nm = subprocess.run(["nm", "-D", "--defined-only", capi.INSTR_LIB_PATH], capture_output=True, text=True).stdout
    for n in list(decl) + dbg:
        assert re.search(rf"\bT {n}\b", nm), f"{n} missing from the instrumented library"
    # no tap code in the product kernels: the instrumented image is the bigger one
    assert os.path.getsize(capi.INSTR_LIB_PATH) > os.path.getsize(capi.LIB_PATH)


def test_frame_geometry_and_state_size():
    L = capi.lib()
    assert L.rnnoise_get_frame_size() == 480  # rnnoise.h:62
    # self-contained POD: header + the 25,128 live bytes of the reference's DenoiseState
    assert 25128 <= L.rnnoise_get_size() <= 25128 + 64


def test_weight_bytes_match_survey_formula(blob_default, blob_little):
    m = capi.Model(blob_default)
    assert m.weight_bytes == 1521668  # SURVEY 8d, measured from the reference's blob
    assert capi.Model(blob_little).weight_bytes < 1521668


def _records(blob):
    off, out = 0, []
    while off < len(blob):
        _, _, typ, size, bs = struct.unpack_from("<4siiii", blob, off)
        name = blob[off + 20:off + 64].split(b"\0")[0].decode()
        out.append((name, off, size, bs))
        off += 64 + bs
    return out


def test_blob_layout_is_the_reference_format(blob_default):
    recs = _records(blob_default)
    assert len(recs) == 43 and len(blob_default) == 1553664
    assert blob_default[:4] == b"DNNw"
    assert all(bs % 64 == 0 and bs >= size for _, _, size, bs in recs)


@pytest.mark.parametrize("damage", ["truncate", "drop_record", "bad_idx", "wrong_size", "empty"])
def test_corrupt_blobs_are_rejected(blob_default, damage):
    b = bytearray(blob_default)
    recs = _records(blob_default)
    if damage == "truncate":
        b = b[: len(b) - 100]
    elif damage == "drop_record":
        name, off, size, bs = next(r for r in recs if r[0] == "gru2_recurrent_scale")
        del b[off: off + 64 + bs]
    elif damage == "bad_idx":
        name, off, size, bs = next(r for r in recs if r[0] == "gru1_input_weights_idx")
        struct.pack_into("<i", b, off + 64 + 4, 382)  # column not a multiple of 4 / out of range
    elif damage == "wrong_size":
        name, off, size, bs = next(r for r in recs if r[0] == "dense_out_bias")
        struct.pack_into("<i", b, off + 12, size - 4)
    elif damage == "empty":
        b = bytearray(64)
    m = capi.Model(bytes(b))
    assert capi.lib().rnnoise_model_weight_bytes(m.h) == -1
    st = (C.c_char * capi.lib().rnnoise_get_size())()
    assert capi.lib().rnnoise_init(C.cast(st, C.c_void_p), m.h) == -1  # rnnoise.h:71 error contract
    assert not capi.lib().rnnoise_create(m.h)


def test_null_arguments_fail_cleanly():
    L = capi.lib()
    assert not L.rnnoise_model_from_buffer(None, 0)
    assert not L.rnnoise_model_from_filename(b"/nonexistent/weights_blob.bin")
    assert not L.rnnoise_create(None)  # no compiled-in model in this build
    assert not L.rnnoise_batch_create(None, 4, 0)
    assert L.rnnoise_batch_size(None) == -1


@pytest.mark.skipif(HAVE_GPU, reason="a GPU is visible")
def test_no_gpu_means_loud_failure_not_cpu_fallback(blob_default, capfd):
    m = capi.Model(blob_default)
    assert capi.lib().rnnoise_amd_device_count() == 0
    with pytest.raises(RuntimeError):
        capi.Batch(m, 4)
    with pytest.raises(RuntimeError):
        capi.DenoiseState(m)
    err = capfd.readouterr().err
    assert "no CPU" in err or "no HIP device" in err


def test_product_never_touches_the_oracle():
    """the oracle is test infrastructure: nothing under rnnoise_amd/ may import, link or open it"""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "rnnoise_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", "Makefile")):
                src = open(os.path.join(dirpath, f), errors="ignore").read()
                code = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
                code = re.sub(r"//[^\n]*", "", code)
                if f.endswith(".py"):
                    code = re.sub(r'\"\"\".*?\"\"\"', "", src, flags=re.S)
                    code = re.sub(r"#[^\n]*", "", code)
                for bad in ("liboracle", "rn_oracle", "oracle.binding", "from oracle", "import oracle", "_ref/", "oracle/"):
                    assert bad not in code, f"{f} references {bad}"
    ldd = subprocess.run(["ldd", capi.LIB_PATH], capture_output=True, text=True).stdout
    assert "oracle" not in ldd and "rnnoise_ref" not in ldd


def test_rcp_profile_selection_is_host_side_and_named():
    """include/rnnoise_amd.h: the profile is process-wide host state (no GPU needed to pick it); "host" captures this CPU's
    rcpps and says which built-in table, if any, it equals"""
    capi.set_rcp_profile("intel")
    assert capi.rcp_profile() == "intel"
    capi.set_rcp_profile("amd")
    assert capi.rcp_profile() == "amd-zen5"
    capi.set_rcp_profile("host")
    name = capi.rcp_profile()
    assert name in ("host=intel", "host=amd-zen5", "host=captured")
    cpu = open("/proc/cpuinfo").read()
    if "GenuineIntel" in cpu:
        assert name == "host=intel"
    if "AMD EPYC 9" in cpu and "9575F" in cpu:
        assert name == "host=amd-zen5"
    with pytest.raises(ValueError):
        capi.set_rcp_profile("m68k")
    assert capi.rcp_profile() == name  # a rejected name changes nothing


def test_c_thread_harness_compiles_against_the_drop_in_header(tmp_path):
    """tools/configs0_mt.c (the pthread program behind profiles/r3_configs0_cthreads.txt) uses nothing but include/rnnoise.h"""
    import subprocess
    obj = tmp_path / "configs0_mt.o"
    subprocess.check_call(["gcc", "-O2", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), "-c",
                           os.path.join(ROOT, "tools", "configs0_mt.c"), "-o", str(obj)])
    syms = subprocess.run(["nm", "-u", str(obj)], capture_output=True, text=True, check=True).stdout
    used = sorted(l.split()[-1] for l in syms.splitlines() if " rnnoise_" in l)
    assert used == ["rnnoise_create", "rnnoise_destroy", "rnnoise_model_free", "rnnoise_model_from_filename", "rnnoise_process_frame"]
