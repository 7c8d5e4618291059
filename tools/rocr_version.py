#!/usr/bin/env python3
"""The ROCm release and the ROCr (HSA runtime) build this box runs: what copy mode "sdma" of the host-fed path was validated on
(rnnoise_amd/csrc/host_io.cpp: sdma_value_word / sdma_selftest)."""
import ctypes as C
import os

print("ROCm release:", open("/opt/rocm/.info/version").read().strip() if os.path.exists("/opt/rocm/.info/version") else "?")
h = C.CDLL("libhsa-runtime64.so.1")
assert h.hsa_init() == 0
major, minor, build = C.c_uint16(), C.c_uint16(), C.c_char_p()
h.hsa_system_get_info(0, C.byref(major))      # HSA_SYSTEM_INFO_VERSION_MAJOR
h.hsa_system_get_info(1, C.byref(minor))      # HSA_SYSTEM_INFO_VERSION_MINOR
rc = h.hsa_system_get_info(0x200, C.byref(build))  # HSA_AMD_SYSTEM_INFO_BUILD_VERSION
freq = C.c_uint64()
h.hsa_system_get_info(3, C.byref(freq))       # HSA_SYSTEM_INFO_TIMESTAMP_FREQUENCY
print(f"HSA interface {major.value}.{minor.value}; ROCr build {build.value.decode() if rc == 0 and build.value else '?'}; system timestamp {freq.value} Hz")
h.hsa_shut_down()
