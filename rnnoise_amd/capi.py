"""ctypes binding of librnnoise_amd.so -- the C-ABI boundary of the product.

Python host-side mirror of the reference's plugin interface for this path: the names,
argument meaning and error behaviour are those of include/rnnoise.h (drop-in, reference
rnnoise.h:51-125) and include/rnnoise_amd.h (additive batched API).  There is no fallback:
if the HIP library is missing or no GPU is visible, construction raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RNNOISE_AMD_LIB", os.path.join(HERE, "librnnoise_amd.so"))  # env override: A/B builds
# the instrumented build of the same sources (-DRN_INSTRUMENT=1): stage taps + probe kernels + include/rnnoise_amd_debug.h
INSTR_LIB_PATH = os.path.join(HERE, "librnnoise_amd_instr.so")

FRAME = 480
NB_BANDS = 32
NB_FEATURES = 65
STATE_FLOATS = 6282

_lib = None
_product = None
_instr = None

# every symbol declared in include/rnnoise.h and include/rnnoise_amd.h
EXPORTS = [
    "rnnoise_get_size", "rnnoise_get_frame_size", "rnnoise_init", "rnnoise_create", "rnnoise_destroy",
    "rnnoise_process_frame", "rnnoise_model_from_buffer", "rnnoise_model_from_file",
    "rnnoise_model_from_filename", "rnnoise_model_free",
    "rnnoise_amd_device_count", "rnnoise_batch_create", "rnnoise_batch_destroy", "rnnoise_batch_size",
    "rnnoise_batch_reset", "rnnoise_batch_process", "rnnoise_batch_process_device",
    "rnnoise_batch_process_s16", "rnnoise_batch_process_device_s16",
    "rnnoise_batch_export_state", "rnnoise_batch_import_state", "rnnoise_batch_set_nn_path",
    "rnnoise_model_weight_bytes", "rnnoise_batch_debug_last", "rnnoise_batch_enable_timing",
    "rnnoise_batch_kernel_ms",
    "rnnoise_batch_train_features", "rnnoise_batch_train_features_device", "rnnoise_amd_model_pack", "rnnoise_batch_set_schedule",
    "rnnoise_amd_set_rcp_profile", "rnnoise_amd_rcp_profile", "rnnoise_amd_log10_model",
]


def _share_hip_runtime_with_torch():
    """One HIP runtime per process.  PyTorch wheels bundle their own libamdhip64.so (SONAME
    libamdhip64.so.7, the same as /opt/rocm's).  If our library were loaded first it would pull
    in the system runtime and a later `import torch` would bring a second one: streams and
    events could then not be shared.  Pre-loading the runtime torch will use makes our
    DT_NEEDED entry resolve to it, whichever import order the application picks."""
    import importlib.util
    spec = importlib.util.find_spec("torch")
    if spec and spec.origin:
        p = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
        if os.path.exists(p):
            C.CDLL(p, mode=C.RTLD_GLOBAL)


# entry points of include/rnnoise_amd_debug.h: present in the instrumented library only
DEBUG_EXPORTS = ["rnnoise_batch_debug_pitch", "rnnoise_amd_debug_fft", "rnnoise_amd_debug_log_energy", "rnnoise_amd_debug_log_energy_range",
                 "rnnoise_amd_debug_gru_race"]


def lib():
    """the library every call of this module goes to: the product, or -- inside `with instrumented():` -- its instrumented twin"""
    global _lib, _product
    if _lib is None:
        if _product is None:
            _product = _load(LIB_PATH, debug=False)
        _lib = _product
    return _lib


class instrumented:
    """`with capi.instrumented():` -- models, batches and calls inside the block use librnnoise_amd_instr.so (taps compiled into
    the kernels, probe kernels, the rnnoise_amd_debug.h entry points).  Objects must not cross the boundary: the two libraries
    are separate images of the same code with separate state.  Tests and tools only; the product never loads it."""

    def __enter__(self):
        global _lib, _instr
        self.prev = lib()
        if _instr is None:
            _instr = _load(INSTR_LIB_PATH, debug=True)
        _lib = _instr
        if self.prev is not _instr:
            _instr.rnnoise_amd_set_rcp_profile(self.prev.rnnoise_amd_rcp_profile().split(b"=")[0])
        return _instr

    def __exit__(self, *exc):
        global _lib
        _lib = self.prev
        return False


def _load(path, debug):
    if True:
        _share_hip_runtime_with_torch()
        if not os.path.exists(path):
            raise RuntimeError(
                f"{path} not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        L = C.CDLL(path)
        vp, fp, ip = C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int)
        L.rnnoise_get_size.restype = C.c_int
        L.rnnoise_get_frame_size.restype = C.c_int
        L.rnnoise_init.argtypes = [vp, vp]
        L.rnnoise_create.restype = vp
        L.rnnoise_create.argtypes = [vp]
        L.rnnoise_destroy.argtypes = [vp]
        L.rnnoise_process_frame.restype = C.c_float
        L.rnnoise_process_frame.argtypes = [vp, fp, fp]
        L.rnnoise_model_from_buffer.restype = vp
        L.rnnoise_model_from_buffer.argtypes = [C.c_char_p, C.c_int]
        L.rnnoise_model_from_filename.restype = vp
        L.rnnoise_model_from_filename.argtypes = [C.c_char_p]
        L.rnnoise_model_free.argtypes = [vp]
        L.rnnoise_amd_device_count.restype = C.c_int
        L.rnnoise_batch_create.restype = vp
        L.rnnoise_batch_create.argtypes = [vp, C.c_int, C.c_int]
        L.rnnoise_batch_destroy.argtypes = [vp]
        L.rnnoise_batch_size.argtypes = [vp]
        L.rnnoise_batch_reset.argtypes = [vp]
        L.rnnoise_batch_process.argtypes = [vp, fp, fp, fp, fp, C.c_int]
        L.rnnoise_batch_process_device.argtypes = [vp, vp, vp, vp, vp, C.c_int, vp]
        L.rnnoise_batch_process_s16.argtypes = [vp, C.POINTER(C.c_short), C.POINTER(C.c_short), fp, fp, C.c_int]
        L.rnnoise_batch_process_device_s16.argtypes = [vp, vp, vp, vp, vp, C.c_int, vp]
        L.rnnoise_batch_export_state.argtypes = [vp, C.c_int, fp]
        L.rnnoise_batch_import_state.argtypes = [vp, C.c_int, fp]
        L.rnnoise_batch_set_nn_path.argtypes = [vp, C.c_int]
        L.rnnoise_batch_set_schedule.argtypes = [vp, C.c_int]
        L.rnnoise_amd_model_pack.restype = C.c_long
        L.rnnoise_amd_model_pack.argtypes = [vp, vp, C.c_long]
        L.rnnoise_model_weight_bytes.restype = C.c_long
        L.rnnoise_model_weight_bytes.argtypes = [vp]
        L.rnnoise_batch_debug_last.argtypes = [vp, fp, ip, ip]
        L.rnnoise_batch_train_features.argtypes = [vp, fp, fp, fp, fp, ip, ip, ip, C.c_int]
        L.rnnoise_batch_train_features_device.argtypes = [vp] * 8 + [C.c_int, vp]
        if debug:
            L.rnnoise_batch_debug_pitch.argtypes = [vp, fp]
            L.rnnoise_amd_debug_log_energy.argtypes = [C.c_int, fp, fp, C.c_int]
            L.rnnoise_amd_debug_log_energy_range.argtypes = [C.c_int, fp, fp, C.c_uint, C.c_uint, C.c_int]
            L.rnnoise_amd_debug_gru_race.argtypes = [C.c_int, C.POINTER(C.c_uint), C.c_int]
            L.rnnoise_amd_debug_fft.argtypes = [C.c_int, C.c_int, fp, fp, C.c_int, C.c_int, C.POINTER(C.c_ulonglong), ip]
        L.rnnoise_batch_enable_timing.argtypes = [vp, C.c_int]
        L.rnnoise_batch_kernel_ms.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_long)]
        L.rnnoise_amd_set_rcp_profile.argtypes = [C.c_char_p]
        L.rnnoise_amd_rcp_profile.restype = C.c_char_p
        L.rnnoise_amd_log10_model.restype = C.c_char_p
    return L


def set_rcp_profile(name: str) -> None:
    """which CPU family's `rcpps` the activations reproduce: "host" (default) | "intel" | "amd-zen5" (include/rnnoise_amd.h)"""
    if lib().rnnoise_amd_set_rcp_profile(name.encode()) != 0:
        raise ValueError(f"unknown rcp profile {name!r}")


def rcp_profile() -> str:
    return lib().rnnoise_amd_rcp_profile().decode()


def log10_model() -> str:
    """which log10 the feature stage evaluates: "host=glibc-fma" (the host libm's algorithm restated on the device) | "glibc-fma" |
    "ocml" | "host=unknown:ocml" (include/rnnoise_amd.h; $RNNOISE_AMD_LOG10)"""
    return lib().rnnoise_amd_log10_model().decode()


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float)) if a is not None else None


def _close_quietly(obj):
    """__del__ helper: at interpreter shutdown this module's globals (and possibly the HIP runtime the handle lives in)
    are already gone -- the process exit reclaims the rest; anywhere else a destructor must not raise."""
    try:
        import sys as _sys
        if _sys is None or _sys.is_finalizing():
            return
        obj.close()
    except Exception:
        pass


class Model:
    """RNNModel from a "DNNw" weight blob (reference: rnnoise_model_from_buffer, rnnoise.h:102)."""

    def __init__(self, blob: bytes):
        self._L = lib()  # the library image that owns this handle (capi.instrumented() swaps the global one)
        self._blob = bytes(blob)  # borrowed by the library for the model's lifetime
        self.h = self._L.rnnoise_model_from_buffer(self._blob, len(self._blob))
        if not self.h:
            raise ValueError("rnnoise_model_from_buffer failed")

    @property
    def weight_bytes(self) -> int:
        w = self._L.rnnoise_model_weight_bytes(self.h)
        if w < 0:
            raise ValueError("weight blob rejected")
        return int(w)

    def pack(self) -> bytes:
        """the model as a GPU-native "RNPK" pack (rnnoise_amd_model_pack); loadable wherever a blob is"""
        n = self._L.rnnoise_amd_model_pack(self.h, None, 0)
        if n <= 0:
            raise ValueError("weight blob rejected")
        buf = C.create_string_buffer(n)
        if self._L.rnnoise_amd_model_pack(self.h, buf, n) != n:
            raise RuntimeError("rnnoise_amd_model_pack failed")
        return buf.raw

    def close(self):
        if getattr(self, "h", None):
            self._L.rnnoise_model_free(self.h)
            self.h = None

    def __del__(self):
        try:
            _close_quietly(self)
        except Exception:  # (at interpreter shutdown the helper itself may already be gone)
            pass


class Batch:
    """N concurrent streams on one GPU (include/rnnoise_amd.h)."""

    def __init__(self, model: Model, n_streams: int, device: int = 0):
        self._L = lib()  # the library image that owns this handle (capi.instrumented() swaps the global one)
        self.model = model
        self.n = n_streams
        self.h = self._L.rnnoise_batch_create(model.h, n_streams, device)
        if not self.h:
            raise RuntimeError("rnnoise_batch_create failed (no GPU / bad model / out of memory)")

    def close(self):
        if getattr(self, "h", None):
            self._L.rnnoise_batch_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            _close_quietly(self)
        except Exception:  # (at interpreter shutdown the helper itself may already be gone)
            pass

    def reset(self):
        if self._L.rnnoise_batch_reset(self.h):
            raise RuntimeError("reset failed")

    def set_nn_path(self, path: int) -> int:
        r = self._L.rnnoise_batch_set_nn_path(self.h, path)
        if r < 0:
            raise RuntimeError(f"network path {path} unsupported")
        return r

    def set_schedule(self, schedule: int) -> int:
        r = self._L.rnnoise_batch_set_schedule(self.h, schedule)
        if r < 0:
            raise RuntimeError(f"schedule {schedule} unsupported")
        return r

    def process(self, pcm: np.ndarray, want_gains: bool = True):
        """pcm: (T, N, 480) float32 host array -> (out, vad[T,N], gains[T,N,32])."""
        pcm = np.ascontiguousarray(pcm, np.float32)
        T, N, F = pcm.shape
        assert N == self.n and F == FRAME
        out = np.empty_like(pcm)
        vad = np.empty((T, N), np.float32)
        gains = np.empty((T, N, NB_BANDS), np.float32) if want_gains else None
        if self._L.rnnoise_batch_process(self.h, _fp(out), _fp(pcm), _fp(vad), _fp(gains), T):
            raise RuntimeError("rnnoise_batch_process failed")
        return out, vad, gains

    def process_s16(self, pcm: np.ndarray, want_gains: bool = True):
        """pcm: (T, N, 480) int16 host array -> (out int16, vad[T,N], gains[T,N,32]): rnnoise_batch_process_s16, the
        conversions of examples/rnnoise_demo.c:56,58 done on the device."""
        pcm = np.ascontiguousarray(pcm, np.int16)
        T, N, F = pcm.shape
        assert N == self.n and F == FRAME
        out = np.empty_like(pcm)
        vad = np.empty((T, N), np.float32)
        gains = np.empty((T, N, NB_BANDS), np.float32) if want_gains else None
        sp = C.POINTER(C.c_short)
        if self._L.rnnoise_batch_process_s16(self.h, out.ctypes.data_as(sp), pcm.ctypes.data_as(sp), _fp(vad), _fp(gains), T):
            raise RuntimeError("rnnoise_batch_process_s16 failed")
        return out, vad, gains

    def process_into(self, out_ptr: int, in_ptr: int, vad_ptr: int, gains_ptr: int, n_frames: int, s16: bool = False):
        """rnnoise_batch_process[_s16] on raw HOST pointers (ints).  Pinned memory (hipHostMalloc / torch pin_memory) is read
        and written by DMA in place; pageable memory goes through the library's pinned bounce buffers."""
        fp = C.POINTER(C.c_float)
        pp = C.POINTER(C.c_short) if s16 else fp
        fn = self._L.rnnoise_batch_process_s16 if s16 else self._L.rnnoise_batch_process
        if fn(self.h, C.cast(out_ptr, pp), C.cast(in_ptr, pp), C.cast(vad_ptr or None, fp), C.cast(gains_ptr or None, fp), n_frames):
            raise RuntimeError("rnnoise_batch_process failed")

    def process_device(self, d_out: int, d_in: int, d_vad: int, d_gains: int, n_frames: int, stream: int = 0, s16: bool = False):
        """Raw device pointers (ints), asynchronous on `stream` (a hipStream_t handle); s16: the PCM buffers hold int16."""
        fn = self._L.rnnoise_batch_process_device_s16 if s16 else self._L.rnnoise_batch_process_device
        if fn(self.h, d_out, d_in, d_vad or None, d_gains or None, n_frames, stream or None):
            raise RuntimeError("rnnoise_batch_process_device failed")

    def export_state(self, stream: int) -> np.ndarray:
        s = np.empty(STATE_FLOATS, np.float32)
        if self._L.rnnoise_batch_export_state(self.h, stream, _fp(s)):
            raise RuntimeError("export_state failed")
        return s

    def import_state(self, stream: int, state: np.ndarray):
        state = np.ascontiguousarray(state, np.float32)
        if self._L.rnnoise_batch_import_state(self.h, stream, _fp(state)):
            raise RuntimeError("import_state failed")

    def debug_last(self):
        f = np.empty((self.n, NB_FEATURES), np.float32)
        s = np.empty(self.n, np.int32)
        p = np.empty(self.n, np.int32)
        ip = C.POINTER(C.c_int)
        if self._L.rnnoise_batch_debug_last(self.h, _fp(f), s.ctypes.data_as(ip), p.ctypes.data_as(ip)):
            raise RuntimeError("debug_last failed")
        return f, s, p

    def train_features(self, clean, noisy, vad, lowpass, band_lp, noise_free):
        """clean/noisy: (T, N, 480); vad: (T, N); lowpass/band_lp/noise_free: (N,) ints -> records (T, N, 98)."""
        clean = np.ascontiguousarray(clean, np.float32)
        noisy = np.ascontiguousarray(noisy, np.float32)
        vad = np.ascontiguousarray(vad, np.float32)
        T, N, _ = clean.shape
        ia = [np.ascontiguousarray(a, np.int32) for a in (lowpass, band_lp, noise_free)]
        rec = np.empty((T, N, 98), np.float32)
        ip = C.POINTER(C.c_int)
        if self._L.rnnoise_batch_train_features(self.h, _fp(rec), _fp(clean), _fp(noisy), _fp(vad),
                                              *[a.ctypes.data_as(ip) for a in ia], T):
            raise RuntimeError("rnnoise_batch_train_features failed")
        return rec

    def debug_pitch(self, arm_only: bool = False):
        if arm_only:
            self._L.rnnoise_batch_debug_pitch(self.h, None)
            return None
        d = np.empty((self.n, 1400), np.float32)
        if self._L.rnnoise_batch_debug_pitch(self.h, _fp(d)):
            raise RuntimeError("debug_pitch failed")
        return d

    def enable_timing(self, on: bool = True):
        self._L.rnnoise_batch_enable_timing(self.h, int(on))

    def kernel_ms(self):
        ms = (C.c_double * 4)()
        n = C.c_long(0)
        if self._L.rnnoise_batch_kernel_ms(self.h, ms, C.byref(n)):
            raise RuntimeError("kernel_ms failed")
        return dict(analysis=ms[0], network=ms[1], synthesis=ms[2], highpass=ms[3], launches=n.value)


class DenoiseState:
    """The reference's one-stream object (rnnoise_create / process_frame / destroy)."""

    def __init__(self, model: Model):
        self._L = lib()  # the library image that owns this handle (capi.instrumented() swaps the global one)
        self.model = model
        self.h = self._L.rnnoise_create(model.h)
        if not self.h:
            raise RuntimeError("rnnoise_create failed")

    def process_frame(self, frame: np.ndarray):
        x = np.ascontiguousarray(frame, np.float32).copy()
        vad = self._L.rnnoise_process_frame(self.h, _fp(x), _fp(x))  # in place, like rnnoise_demo.c:57
        return x, vad

    def close(self):
        if getattr(self, "h", None):
            self._L.rnnoise_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            _close_quietly(self)
        except Exception:  # (at interpreter shutdown the helper itself may already be gone)
            pass
