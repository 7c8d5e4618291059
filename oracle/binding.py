"""ctypes bindings for the oracle kit.  TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module; the product package (rnnoise_amd/) never does.

  Oracle     -- liboracle.so, our plain-C restatement (oracle/rn_oracle.c); portable.
  RefHarness -- oracle/_ref/libref_harness.so, the UNMODIFIED reference compiled from
                /root/reference plus recording wrappers (oracle/ref_harness.c).  Exists
                only where `make -C oracle ref` has run (needs /root/reference).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
STATE_FLOATS = 6282
NB_BANDS = 32
NB_FEATURES = 65
FRAME = 480


class Record(C.Structure):
    _fields_ = [
        ("features", C.c_float * NB_FEATURES),
        ("gains", C.c_float * NB_BANDS),
        ("vad", C.c_float),
        ("pitch_gain", C.c_float),
        ("pitch", C.c_int),
        ("silence", C.c_int),
    ]


def _fp(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def build_oracle() -> str:
    """Compile liboracle.so if missing or stale (gcc only; works on the GPU box too)."""
    so = os.path.join(HERE, "liboracle.so")
    srcs = [os.path.join(HERE, f) for f in ("rn_oracle.c", "rn_oracle.h", "../rnnoise_amd/csrc/rcp_profiles.h", "../rnnoise_amd/csrc/rcp_profile_intel.h",
                                              "../rnnoise_amd/csrc/rcp_profile_amd_zen5.h")]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def set_rcp_profile(name: str) -> None:
    """which CPU family's `rcpps` the ORACLE's tanh / sigmoid use (rnnoise_amd/csrc/rcp_profiles.h): "intel" (default:
    the committed goldens), "amd-zen5", or "host" = captured from the CPU this process runs on, which is what a live
    comparison against the compiled reference (oracle/_ref) or the product's default profile needs"""
    if Oracle.lib().rno_set_rcp_profile(name.encode()) != 0:
        raise ValueError(f"oracle: unknown or uncapturable rcp profile {name!r}")


def rcp_profile() -> str:
    return ("intel", "amd-zen5", "other")[Oracle.lib().rno_rcp_profile_id()]


class _FrameRunner:
    """Common per-stream driver: process frames, collect records."""

    def run(self, pcm_frames: np.ndarray):
        """pcm_frames: (T, 480) float32.  Returns dict of arrays over T."""
        T = pcm_frames.shape[0]
        out = np.zeros((T, FRAME), np.float32)
        vad = np.zeros(T, np.float32)
        gains = np.zeros((T, NB_BANDS), np.float32)
        feats = np.zeros((T, NB_FEATURES), np.float32)
        pitch = np.zeros(T, np.int32)
        pgain = np.zeros(T, np.float32)
        silence = np.zeros(T, np.int32)
        for t in range(T):
            o, v, rec = self.process(pcm_frames[t])
            out[t] = o
            vad[t] = v
            gains[t] = np.frombuffer(rec.gains, np.float32)
            feats[t] = np.frombuffer(rec.features, np.float32)
            pitch[t] = rec.pitch
            pgain[t] = rec.pitch_gain
            silence[t] = rec.silence
        return dict(out=out, vad=vad, gains=gains, features=feats, pitch=pitch, pitch_gain=pgain, silence=silence)


class Oracle(_FrameRunner):
    _lib = None

    @classmethod
    def lib(cls):
        if cls._lib is None:
            L = C.CDLL(build_oracle())
            L.rno_model_from_blob.restype = C.c_void_p
            L.rno_model_from_blob.argtypes = [C.c_char_p, C.c_int]
            L.rno_model_free.argtypes = [C.c_void_p]
            L.rno_state_init.argtypes = [C.POINTER(C.c_float)]
            L.rno_process_frame.restype = C.c_float
            L.rno_process_frame.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float),
                                            C.POINTER(C.c_float), C.POINTER(Record)]
            L.rno_fft.argtypes = [C.POINTER(C.c_float)] * 2
            L.rno_tables.argtypes = [C.POINTER(C.c_float)] * 3 + [C.POINTER(C.c_int)]
            L.rno_pitch.restype = C.c_float
            L.rno_pitch.argtypes = [C.POINTER(C.c_float), C.c_int, C.c_float, C.POINTER(C.c_int), C.POINTER(C.c_float)]
            L.rno_pitch_debug.restype = C.c_float
            L.rno_pitch_debug.argtypes = [C.POINTER(C.c_float), C.c_int, C.c_float, C.POINTER(C.c_int), C.POINTER(C.c_float)]
            L.rno_train_frame.argtypes = [C.POINTER(C.c_float)] * 4 + [C.c_int, C.c_int, C.c_float, C.c_int, C.POINTER(C.c_float)]
            L.rno_compute_rnn.argtypes = [C.c_void_p] + [C.POINTER(C.c_float)] * 4
            L.rno_band_energy.argtypes = [C.POINTER(C.c_float)] * 2
            L.rno_interp_band_gain.argtypes = [C.POINTER(C.c_float)] * 2
            for f in ("rno_rcp", "rno_tanh", "rno_sigmoid"):
                getattr(L, f).restype = C.c_float
                getattr(L, f).argtypes = [C.c_float]
            L.rno_set_rcp_profile.argtypes = [C.c_char_p]
            L.rno_quantize_u8.argtypes = [C.POINTER(C.c_ubyte), C.POINTER(C.c_float), C.c_int]
            L.rno_log_energy.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int]
            L.rno_log_energy_range_diff.restype = C.c_uint
            L.rno_log_energy_range_diff.argtypes = [C.c_uint, C.c_uint, C.POINTER(C.c_float), C.POINTER(C.c_uint)]
            cls._lib = L
        return cls._lib

    def __init__(self, blob: bytes):
        self.L = self.lib()
        self.model = self.L.rno_model_from_blob(blob, len(blob))
        if not self.model:
            raise ValueError("oracle: blob rejected")
        self.state = np.zeros(STATE_FLOATS, np.float32)

    def __del__(self):
        if getattr(self, "model", None):
            self.L.rno_model_free(self.model)
            self.model = None

    def process(self, frame: np.ndarray):
        frame = np.ascontiguousarray(frame, np.float32)
        out = np.zeros(FRAME, np.float32)
        rec = Record()
        v = self.L.rno_process_frame(self.model, _fp(self.state), _fp(out), _fp(frame), C.byref(rec))
        return out, v, rec

    def compute_rnn(self, features: np.ndarray):
        features = np.ascontiguousarray(features, np.float32)
        g = np.zeros(NB_BANDS, np.float32)
        v = np.zeros(1, np.float32)
        self.L.rno_compute_rnn(self.model, _fp(self.state), _fp(g), _fp(v), _fp(features))
        return g, float(v[0])

    def get_state(self):
        return self.state.copy()

    def set_state(self, s):
        self.state[:] = s

    # stage helpers -----------------------------------------------------------------------
    @classmethod
    def log_energy(cls, ex: np.ndarray) -> np.ndarray:
        """(float)log10(1e-2 + (double)ex) with the host libm (src/denoise.c:383)"""
        ex = np.ascontiguousarray(ex, np.float32)
        out = np.empty_like(ex)
        cls.lib().rno_log_energy(_fp(out), _fp(ex), ex.size)
        return out

    @classmethod
    def log_energy_range_diff(cls, first_bits: int, got: np.ndarray):
        """how many of got[i] differ from (float)log10(1e-2 + (double)float_with_bits(first_bits + i)) under the host libm, and
        the first such bit pattern (or None)"""
        got = np.ascontiguousarray(got, np.float32)
        first = C.c_uint(0)
        n = cls.lib().rno_log_energy_range_diff(first_bits, got.size, _fp(got), C.byref(first))
        return int(n), (int(first.value) if n else None)

    @classmethod
    def fft(cls, x_ri: np.ndarray) -> np.ndarray:
        x_ri = np.ascontiguousarray(x_ri, np.float32)
        y = np.zeros(1920, np.float32)
        cls.lib().rno_fft(_fp(x_ri), _fp(y))
        return y

    @classmethod
    def tables(cls):
        w = np.zeros(480, np.float32); d = np.zeros(1024, np.float32); tw = np.zeros(1920, np.float32)
        br = np.zeros(960, np.int32)
        cls.lib().rno_tables(_fp(w), _fp(d), _fp(tw), br.ctypes.data_as(C.POINTER(C.c_int)))
        return w, d, tw, br

    @classmethod
    def pitch(cls, buf1728: np.ndarray, last_period: int, last_gain: float):
        buf = np.ascontiguousarray(buf1728, np.float32)
        T = C.c_int(0)
        lp = np.zeros(864, np.float32)
        g = cls.lib().rno_pitch(_fp(buf), last_period, last_gain, C.byref(T), _fp(lp))
        return T.value, g, lp


    @classmethod
    def pitch_debug(cls, buf1728: np.ndarray, last_period: int, last_gain: float):
        buf = np.ascontiguousarray(buf1728, np.float32)
        T = C.c_int(0)
        dbg = np.zeros(1400, np.float32)
        g = cls.lib().rno_pitch_debug(_fp(buf), last_period, last_gain, C.byref(T), _fp(dbg))
        return T.value, g, dbg


class RefHarness(_FrameRunner):
    PATH = os.path.join(HERE, "_ref", "libref_harness.so")
    _lib = None

    @classmethod
    def available(cls) -> bool:
        return os.path.exists(cls.PATH)

    @classmethod
    def lib(cls):
        if cls._lib is None:
            L = C.CDLL(cls.PATH)
            L.refh_create.restype = C.c_void_p
            L.refh_create.argtypes = [C.c_char_p, C.c_int]
            L.refh_destroy.argtypes = [C.c_void_p]
            L.refh_arch.argtypes = [C.c_void_p]
            L.refh_process.restype = C.c_float
            L.refh_process.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(Record)]
            L.refh_get_state.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
            L.refh_set_state.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
            L.refh_fft.argtypes = [C.POINTER(C.c_float)] * 2
            L.refh_tables.argtypes = [C.POINTER(C.c_float)] * 3 + [C.POINTER(C.c_int)]
            L.refh_pitch.restype = C.c_float
            L.refh_pitch.argtypes = [C.POINTER(C.c_float), C.c_int, C.c_float, C.POINTER(C.c_int), C.POINTER(C.c_float)]
            cls._lib = L
        return cls._lib

    def __init__(self, blob: bytes | None = None):
        self.L = self.lib()
        self._blob = blob  # the reference borrows the buffer (rnnoise.h:99-100)
        self.h = self.L.refh_create(blob, len(blob) if blob else 0)
        if not self.h:
            raise ValueError("reference rejected the blob")

    def __del__(self):
        if getattr(self, "h", None):
            self.L.refh_destroy(self.h)
            self.h = None

    def process(self, frame: np.ndarray):
        frame = np.ascontiguousarray(frame, np.float32)
        out = np.zeros(FRAME, np.float32)
        rec = Record()
        v = self.L.refh_process(self.h, _fp(out), _fp(frame), C.byref(rec))
        return out, v, rec

    def get_state(self):
        s = np.zeros(STATE_FLOATS, np.float32)
        self.L.refh_get_state(self.h, _fp(s))
        return s

    def set_state(self, s):
        s = np.ascontiguousarray(s, np.float32)
        self.L.refh_set_state(self.h, _fp(s))

    @classmethod
    def fft(cls, x_ri):
        x_ri = np.ascontiguousarray(x_ri, np.float32)
        y = np.zeros(1920, np.float32)
        cls.lib().refh_fft(_fp(x_ri), _fp(y))
        return y

    @classmethod
    def tables(cls):
        w = np.zeros(480, np.float32); d = np.zeros(1024, np.float32); tw = np.zeros(1920, np.float32)
        br = np.zeros(960, np.int32)
        cls.lib().refh_tables(_fp(w), _fp(d), _fp(tw), br.ctypes.data_as(C.POINTER(C.c_int)))
        return w, d, tw, br

    @classmethod
    def pitch(cls, buf1728, last_period, last_gain):
        buf = np.ascontiguousarray(buf1728, np.float32)
        T = C.c_int(0)
        lp = np.zeros(864, np.float32)
        g = cls.lib().refh_pitch(_fp(buf), last_period, last_gain, C.byref(T), _fp(lp))
        return T.value, g, lp


class TrainOracle:
    """oracle side of the training-feature extraction step (rno_train_frame)"""

    def __init__(self):
        self.L = Oracle.lib()
        self.state = np.zeros(STATE_FLOATS, np.float32)
        self.clean_mem = np.zeros(FRAME, np.float32)

    def frame(self, clean, noisy, lowpass=481, band_lp=32, vad=0.0, noise_free=0):
        clean = np.ascontiguousarray(clean, np.float32)
        noisy = np.ascontiguousarray(noisy, np.float32)
        rec = np.zeros(98, np.float32)
        self.L.rno_train_frame(_fp(self.state), _fp(self.clean_mem), _fp(clean), _fp(noisy), lowpass, band_lp, vad,
                               noise_free, _fp(rec))
        return rec


class RefTrainHarness:
    """the reference's own TRAINING=1 functions (oracle/ref_harness_train.c)"""
    PATH = os.path.join(HERE, "_ref", "libref_harness_train.so")

    @classmethod
    def available(cls):
        return os.path.exists(cls.PATH)

    def __init__(self):
        L = C.CDLL(self.PATH)
        L.refht_create.restype = C.c_void_p
        L.refht_destroy.argtypes = [C.c_void_p]
        L.refht_frame.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int, C.c_int, C.c_float,
                                  C.c_int, C.POINTER(C.c_float)]
        self.L = L
        self.h = L.refht_create()

    def __del__(self):
        if getattr(self, "h", None):
            self.L.refht_destroy(self.h)
            self.h = None

    def frame(self, clean, noisy, lowpass=481, band_lp=32, vad=0.0, noise_free=0):
        clean = np.ascontiguousarray(clean, np.float32)
        noisy = np.ascontiguousarray(noisy, np.float32)
        rec = np.zeros(98, np.float32)
        self.L.refht_frame(self.h, _fp(clean), _fp(noisy), lowpass, band_lp, vad, noise_free, _fp(rec))
        return rec
