#!/usr/bin/env python3
"""Race hunt, second instrument: WHICH stage of a frame first differs between replicas.

N streams = replicas of a 32-stream block through pipelined calls of 5 + 1 + 8 frames on the INSTRUMENTED library with the
pitch-stage taps armed; after every repetition the taps of the LAST frame (frame 13) are compared between replicas, stage by
stage in pipeline order:
  ac        K0's five autocorrelation lags (written by rn_hp_kernel from its own registers)
  lpc       the FIR taps as K1 READ them from lpc2[slot] (written by rn_analysis_kernel)
  xlp       K1's decimated + whitened signal; xc_coarse / best / xc_fine / dots: the later pitch stages
and a wrong `lpc` vector is compared with the taps frame 7 left in the same lpc2 slot (13 % 6 == 7 % 6): a STALE read.
The switches under test come in through the environment (RNNOISE_AMD_GRU_VARIANT, RNNOISE_AMD_PIPE, RNNOISE_AMD_HP_AB ...).

usage: RNNOISE_AMD_GRU_VARIANT=w4 RNNOISE_AMD_PIPE=1 tools/race_taps.py [--streams 32768] [--reps 6]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from rnnoise_amd import capi, synth
from conftest import load_blob

STAGES = (("ac (K0)", 864, 5), ("lpc as read by K1", 869, 5), ("xlp", 0, 864), ("xc_coarse", 880, 147), ("best", 1030, 6),
          ("xc_fine", 1040, 294), ("dots", 1340, 7))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=32768); ap.add_argument("--model", default="little"); ap.add_argument("--reps", type=int, default=6)
    a = ap.parse_args()
    N = a.streams
    env = {k: v for k, v in os.environ.items() if k.startswith("RNNOISE_AMD_")}
    print(f"# tools/race_taps.py: {N} streams, env {env}")
    with capi.instrumented():
        capi.set_rcp_profile("intel")
        T = 14
        base = synth.batch_pcm(range(32), T); base[:3, 9] = 0; base[T - 5:T - 3, 12] = 0
        dev = torch.device("cuda", 0)
        d_in = torch.from_numpy(base).to(dev).repeat(1, N // 32, 1).contiguous()
        d_out = torch.empty_like(d_in); d_vad = torch.empty((T, N), device=dev); d_gains = torch.empty((T, N, 32), device=dev)
        m = capi.Model(load_blob(a.model)); b = capi.Batch(m, N); b.set_nn_path(1)
        b.debug_pitch(arm_only=True)
        st = torch.cuda.current_stream().cuda_stream

        def run(calls):
            b.reset(); f = 0
            for n in calls:
                b.process_device(d_out[f].data_ptr(), d_in[f].data_ptr(), d_vad[f].data_ptr(), d_gains[f].data_ptr(), n, st); f += n
            torch.cuda.synchronize()
            return b.debug_pitch().view(np.uint32).reshape(N // 32, 32, 1400)

        old = b.set_schedule(9)
        t7 = run((5, 1, 2))[0].copy()      # frame 7's taps, one-stream schedule
        ref = run((5, 1, 8)).copy()        # frame 13's
        b.set_schedule(old)
        for name, off, ln in STAGES:  # (the clock taps differ, of course)
            if not (ref[:, :, off:off + ln] == ref[:1, :, off:off + ln]).all():
                rp, ss, ww = np.nonzero(ref[:, :, off:off + ln] != ref[:1, :, off:off + ln])
                print(f"# one-stream run: replicas differ in {name}: {len(rp)} words, e.g. replica {rp[0]} stream {ss[0]} word {ww[0]}; words {sorted(set(ww.tolist()))[:10]}")
        ref = ref[0]
        for rep in range(a.reps):
            d = run((5, 1, 8))
            gains_bad = (d_gains.view(torch.int32).reshape(T, N // 32, 32, 32) != d_gains.view(torch.int32).reshape(T, N // 32, 32, 32)[:, :1]).any(dim=3)
            fr = gains_bad.any(dim=2).any(dim=1).nonzero().flatten().tolist()
            line = [f"rep {rep}: gains differ in frames {fr[:1]}..{fr[-1:]} ({int(gains_bad.any(dim=0).sum())} replica-streams)"]
            for name, off, ln in STAGES:
                bad = (d[:, :, off:off + ln] != ref[None, :, off:off + ln]).any(axis=2)  # [replica][stream]
                if not bad.any():
                    line.append(f"  {name}: all {N} streams = the one-stream run")
                    continue
                rp, ss = np.nonzero(bad)
                lanes = sorted(set(((rp % 2) * 32 + ss).tolist()))
                line.append(f"  {name}: {len(rp)} streams differ; replicas {sorted(set(rp.tolist()))[:8]} (odd: {int((rp % 2 == 1).sum())} of {len(rp)}); "
                            f"position in the 64-stream group: {lanes[0]}..{lanes[-1]} ({len(lanes)} distinct)")
                if name.startswith("lpc"):
                    stale = sum(bool((d[r, s, off:off + ln] == t7[s, off:off + ln]).all()) for r, s in zip(rp, ss))
                    line.append(f"      of them equal to frame 7's taps (what the slot held six frames earlier): {stale}")
                    r, s = int(rp[0]), int(ss[0])
                    line.append(f"      e.g. stream {32 * r + s}: read {d[r, s, off:off + ln].view(np.float32).tolist()}\n"
                                f"           frame 13 wants {ref[s, off:off + ln].view(np.float32).tolist()}\n"
                                f"           frame 7 left   {t7[s, off:off + ln].view(np.float32).tolist()}")
                if name.startswith("ac"):
                    r, s = int(rp[0]), int(ss[0])
                    line.append(f"      e.g. stream {32 * r + s}: got {d[r, s, off:off + ln].view(np.float32).tolist()} want {ref[s, off:off + ln].view(np.float32).tolist()}")
            print("\n".join(line), flush=True)
        b.close(); m.close()


if __name__ == "__main__":
    main()
