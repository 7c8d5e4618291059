#!/usr/bin/env python3
"""What ONE float feature rounded the other way does downstream (Weak #2) -- measured on the oracle, CPU only.

For several streams and bands: the stream is run twice over T frames, the second time with Ly[band] of frame F moved to the
adjacent float (what a log10 that rounds a double-rounding tie differently would produce, src/denoise.c:383).  Reported: the
largest relative deviation of the raw gains, for how many frames the gains differ, and whether the state ever re-converges.
The u8 re-quantisers either swallow the 1-ULP feature change at once -- nothing differs -- or turn it into a
1/127 step of one activation, which the GRU state then carries.

    python tools/log10_flip_effect.py > profiles/r5_log10_flip.txt
"""
import lzma
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import binding  # noqa: E402
from oracle.binding import Oracle  # noqa: E402
from rnnoise_amd import synth  # noqa: E402

T, F = 400, 60


def main():
    blob = lzma.decompress(open(os.path.join(ROOT, "tests", "golden", "default.blob.xz"), "rb").read())
    L = Oracle.lib()
    streams = [3, 17, 77, 130, 201, 255]
    bands = [0, 3, 8, 15, 22, 31]
    print(f"# one feature of frame {F} moved by one float ULP (Ly[band] -> nextafter), {T} frames per run, default model, rcp profile "
          f"{binding.rcp_profile()}")
    print("# stream band | frames whose gains differ | first..last such frame | max relative gain deviation | max |d vad| | state equal at the end")
    n_runs = n_hit = 0
    worst = 0.0
    for sid in streams:
        pcm = synth.batch_pcm([sid], T, lead_silence=2)[:, 0]
        base = Oracle(blob).run(pcm)
        for band in bands:
            o = Oracle(blob)
            L.rno_flip_log_energy(F + 1, band)
            got = o.run(pcm)
            L.rno_flip_log_energy(0, -1)
            d = got["gains"].view(np.uint32) != base["gains"].view(np.uint32)
            frames = np.nonzero(d.any(axis=1))[0]
            rel = np.abs(got["gains"] - base["gains"]) / np.maximum(np.abs(base["gains"]), 1e-30)
            dv = np.abs(got["vad"] - base["vad"]).max()
            n_runs += 1
            if frames.size:
                n_hit += 1
                worst = max(worst, float(rel.max()))
                tail = not d[-1].any()
                print(f"{sid:6d} {band:4d} | {frames.size:4d} | {frames[0]:3d}..{frames[-1]:3d} | {rel.max():.3e} | {dv:.3e} | "
                      f"{'gains equal again from frame %d' % (frames[-1] + 1) if tail else 'still different at frame %d' % (T - 1)}")
            else:
                feat_changed = not np.array_equal(got["features"].view(np.uint32), base["features"].view(np.uint32))
                print(f"{sid:6d} {band:4d} |    0 | - | 0 | 0 | features of frame {F} {'differ' if feat_changed else 'identical (follower clamp)'}; "
                      f"swallowed by the quantisers")
    print(f"# {n_hit} of {n_runs} flips reached the gains; worst relative gain deviation {worst:.3e} (north_star's bar: 1e-4)")
    # the pass rate of the quantisers, from MANY flips at once: every band of every frame from F on rounded the other way
    print(f"#\n# every band of every frame from frame {F} on moved by one ULP ({32 * (T - F)} flips per run): frames until the gains first differ")
    print("# stream | first frame whose gains differ | frames that differ | max relative gain deviation")
    flips = firsts = 0
    for sid in streams + [9, 41, 99, 150, 180, 222]:
        pcm = synth.batch_pcm([sid], T, lead_silence=2)[:, 0]
        base = Oracle(blob).run(pcm)
        o = Oracle(blob)
        L.rno_flip_log_energy(F + 1, -2)
        got = o.run(pcm)
        L.rno_flip_log_energy(0, -1)
        d = (got["gains"].view(np.uint32) != base["gains"].view(np.uint32)).any(axis=1)
        frames = np.nonzero(d)[0]
        rel = np.abs(got["gains"] - base["gains"]) / np.maximum(np.abs(base["gains"]), 1e-30)
        live = int((base["silence"][F:(frames[0] if frames.size else T)] == 0).sum())
        flips += 32 * live
        firsts += 1 if frames.size else 0
        print(f"{sid:6d} | {frames[0] if frames.size else '-'} | {frames.size} | {rel.max():.3e}")
    print(f"# {firsts} first deviations in {flips} flipped features of non-silent frames: about one flipped feature in {flips // max(firsts, 1)} reaches "
          f"the gains; once it does, the GRU state carries it (the deviation does not die out) and single gains move by far more than 1e-4")


if __name__ == "__main__":
    main()
