"""PyTorch-ROCm binding of the batched op: forward only, tensors in, tensors out,
on torch's current stream; plus an un-quantised float re-statement of the network built from the
same blob for sanity cross-checks (not bit parity: exact tanh/sigmoid, no 8-bit activations), the
role torch/rnnoise/rnnoise.py:86-109 plays in the reference's training stack.

torch is plumbing here (HBM buffers, streams); the arithmetic runs in librnnoise_amd.so.
"""
from __future__ import annotations

import weakref

import numpy as np

from . import blob as rblob
from . import capi


# handle -> RNNoiseOp: what torch.ops.rnnoise_amd.process resolves its integer argument to.  Weak: an op nobody holds any
# more is closed by its destructor (it owns a GPU arena and a model), not kept alive by this table.
_OPS = weakref.WeakValueDictionary()
_registered = False


def register_torch_op():
    """Registers `torch.ops.rnnoise_amd.process(pcm, state, handle) -> (out, vad, gains)` (torch.library custom op, forward
    only).  `handle` is RNNoiseOp.handle: the op is stateful per stream batch, like the C API it binds, and the stream state
    lives in the library, not in tensors.  So that tracing / functionalization / torch.compile cannot treat it as a pure
    function -- merge two calls with the same arguments, or drop one whose outputs are unused, and silently desynchronise the
    streams -- the op MUTATES its `state` argument (RNNoiseOp.state: the batch's frame counter, a one-element int64 tensor,
    advanced by the number of frames of every call).  A fake (meta) implementation gives shapes."""
    global _registered
    if _registered:
        return
    import torch

    # (explicit schema: this module uses postponed annotations, which infer_schema cannot resolve for a local import)
    @torch.library.custom_op("rnnoise_amd::process", mutates_args=("state",),
                             schema="(Tensor pcm, Tensor(a!) state, int handle) -> (Tensor, Tensor, Tensor)")
    def process(pcm, state, handle):
        res = _OPS[handle]._run(pcm)
        state.add_(pcm.shape[0])
        return res

    @process.register_fake
    def _(pcm, state, handle):
        T, N = pcm.shape[0], pcm.shape[1]
        return torch.empty_like(pcm), pcm.new_empty((T, N)), pcm.new_empty((T, N, capi.NB_BANDS))

    _registered = True


class RNNoiseOp:
    """N concurrent streams; call with a (T, N, 480) float32 CUDA tensor of int16-scaled PCM.  The same object is
    reachable as the registered op: torch.ops.rnnoise_amd.process(pcm, op.state, op.handle)."""

    def __init__(self, model_blob: bytes, n_streams: int, device: int = 0, nn_path: str = "mfma"):
        import torch
        self.torch = torch
        self.device = torch.device("cuda", device)
        self.model = capi.Model(model_blob)
        self.batch = capi.Batch(self.model, n_streams, device=device)
        if nn_path == "mfma":
            self.batch.set_nn_path(1)
        self.n = n_streams
        register_torch_op()
        self.handle = id(self)
        self.state = torch.zeros(1, dtype=torch.int64, device=self.device)  # frames processed: the tensor the op mutates
        _OPS[self.handle] = self

    def close(self):
        _OPS.pop(getattr(self, "handle", None), None)
        if getattr(self, "batch", None) is not None:
            self.batch.close()
            self.model.close()
            self.batch = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __call__(self, pcm):
        return self.torch.ops.rnnoise_amd.process(pcm, self.state, self.handle)

    def _run(self, pcm):
        torch = self.torch
        assert pcm.is_cuda and pcm.dtype == torch.float32 and pcm.shape[1:] == (self.n, capi.FRAME)
        pcm = pcm.contiguous()
        T = pcm.shape[0]
        out = torch.empty_like(pcm)
        vad = torch.empty((T, self.n), device=pcm.device, dtype=torch.float32)
        gains = torch.empty((T, self.n, capi.NB_BANDS), device=pcm.device, dtype=torch.float32)
        self.batch.process_device(out.data_ptr(), pcm.data_ptr(), vad.data_ptr(), gains.data_ptr(), T,
                                  torch.cuda.current_stream(pcm.device).cuda_stream)
        return out, vad, gains

    def reset(self):
        self.batch.reset()
        self.state.zero_()


class FloatNet:
    """Float32 forward of the network from a blob: int8 layers de-quantised (w_q * scale * 127), exact
    tanh/sigmoid, float activations.  State handling as src/rnn.c:44-60 / src/nnet.c:65-123."""

    def __init__(self, model_blob: bytes):
        r = rblob.read_blob(model_blob)
        f = lambda a: np.asarray(a, np.float64)  # noqa: E731
        self.c1w, self.c1b = f(r["conv1_weights_float"]).reshape(195, 128), f(r["conv1_bias"])
        q = f(r["conv2_weights_int8"]).reshape(48, 96, 8, 4).transpose(1, 3, 0, 2).reshape(384, 384)  # -> (in, out)
        self.c2w, self.c2b = q * f(r["conv2_scale"]) * 127, f(r["conv2_bias"])
        self.gru = []
        for k in (1, 2, 3):
            mats = []
            for side in ("input", "recurrent"):
                name = f"gru{k}_{side}"
                w = np.zeros((384, 1152))
                idx, blocks = r[name + "_weights_idx"], f(r[name + "_weights_int8"]).reshape(-1, 8, 4)
                p = b = 0
                for grp in range(144):
                    nb = idx[p]; p += 1
                    for _ in range(nb):
                        col = idx[p]; p += 1
                        w[col:col + 4, 8 * grp:8 * grp + 8] = blocks[b].T
                        b += 1
                w = w * f(r[name + "_scale"]) * 127
                if side == "recurrent":
                    d = f(r[name + "_weights_diag"])
                    for g in range(3):
                        w[np.arange(384), g * 384 + np.arange(384)] += d[g * 384:(g + 1) * 384]
                mats.append((w, f(r[name + "_bias"])))
            self.gru.append(mats)
        self.dw, self.db = f(r["dense_out_weights_float"]).reshape(1536, 32), f(r["dense_out_bias"])
        self.vw, self.vb = f(r["vad_dense_weights_float"]).reshape(1536, 1), f(r["vad_dense_bias"])
        self.reset()

    def reset(self):
        self.c1s, self.c2s, self.h = np.zeros(130), np.zeros(256), [np.zeros(384) for _ in range(3)]

    def step(self, features65):
        sig = lambda x: 1 / (1 + np.exp(-x))  # noqa: E731
        t1 = np.concatenate([self.c1s, features65])
        c1 = np.tanh(t1 @ self.c1w + self.c1b)
        self.c1s = t1[65:]
        t2 = np.concatenate([self.c2s, c1])
        x = np.tanh(t2 @ self.c2w + self.c2b)
        self.c2s = t2[128:]
        cat = [x]
        for k, ((wi, bi), (wr, br)) in enumerate(self.gru):
            a, b = x @ wi + bi, self.h[k] @ wr + br
            z, rr = sig(a[:384] + b[:384]), sig(a[384:768] + b[384:768])
            hh = np.tanh(a[768:] + b[768:] * rr)
            self.h[k] = z * self.h[k] + (1 - z) * hh
            x = self.h[k]
            cat.append(x)
        cat = np.concatenate(cat)
        return sig(cat @ self.dw + self.db), float(sig(cat @ self.vw + self.vb)[0])
