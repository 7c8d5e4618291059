#!/usr/bin/env python3
"""Lane-level emulation (numpy, float32 = IEEE, so bit-faithful) of the register-resident 960-point FFT used by the
analysis / synthesis kernels, checked bit for bit against the oracle's kiss_fft restatement (oracle/rn_oracle.c: fft960).

Layout: lane l (0..63) holds 15 complex values a[blk], blk = 0..14, the FFT work-area positions p = 64*blk + P(l) where
P(l) starts as l and, after each of the three cross-lane radix-4 stages, has the stage's 2-bit digit bit-reversed
(roles 1 and 2 of a butterfly end up swapped; the next stages' twiddles are simply indexed with the true position).
Input: work-area position p holds natural sample i with bitrev(i) = p; for p = 64*blk + l that is
i = 15*lam(l) + c(blk), lam(l) = base-4 digit reversal of l, c(blk) = blk//3 + 5*(blk%3): 15 CONSECUTIVE samples per lane.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
f32 = np.float32
L = np.arange(64)


def lam(l):  # base-4 digit reversal of the lane number: l = 16*j2 + 4*j3 + j4 -> j2 + 4*j3 + 16*j4
    return (l >> 4) + 4 * ((l >> 2) & 3) + 16 * (l & 3)


SIG = np.array([0, 2, 1, 3])


def c_of_blk(blk):
    return blk // 3 + 5 * (blk % 3)


def twiddles():
    i = np.arange(960)
    ph = (-2 * np.pi / 960) * i  # computed in double, rounded to float (kiss_fft.c:415-419)
    pi = 3.14159265358979323846264338327
    ph = (-2 * pi / 960) * i
    return np.cos(ph).astype(f32), np.sin(ph).astype(f32)


TWR, TWI = twiddles()


def cmul(ar, ai, br, bi):  # src/_kiss_fft_guts.h:101-103, unfused
    return (ar * br - ai * bi).astype(f32), (ar * bi + ai * br).astype(f32)


def radix4_stage(ar, ai, shift, pos_low, fstride):
    """one cross-lane radix-4 stage on digit (l >> shift) & 3.  ar/ai: [15][64].  pos_low[l] = true position of the
    lane's element modulo m (the butterfly's j), m = 4**(shift/2); fstride = twiddle stride of the stage (0: none)."""
    k = (L >> shift) & 3
    if fstride:
        e = (fstride * pos_low * k) % 960
        tr, ti = cmul(ar, ai, TWR[e][None, :], TWI[e][None, :])
        ar = np.where(k[None, :] == 0, ar, tr)  # role 0 is not multiplied in the reference (keeps signed zeros)
        ai = np.where(k[None, :] == 0, ai, ti)
    p2 = L ^ (2 << shift)
    p1 = L ^ (1 << shift)
    s1 = np.where(k < 2, f32(1), f32(-1))[None, :]
    tr = (ar[:, p2] + s1 * ar).astype(f32)  # level 1: E+ (k=0), O+ (k=1), E- (k=2), O- (k=3)
    ti = (ai[:, p2] + s1 * ai).astype(f32)
    is3 = (k == 3)[None, :]
    wr = np.where(is3, ti, tr)              # rot(t) = (t.i, -t.r) on role 3
    wi = np.where(is3, -tr, ti)
    s2 = np.where((k & 1) == 0, f32(1), f32(-1))[None, :]
    outr = (wr[:, p1] + s2 * wr).astype(f32)  # out0 (k=0), out2 (k=1), out1 (k=2), out3 (k=3)
    outi = (wi[:, p1] + s2 * wi).astype(f32)
    return outr, outi


def regfft(x_r, x_i):
    """x: natural-order input (960), ALREADY scaled like the reference (x * 1/960 applied by the caller).
    returns (yr, yi) [15][64] and pos[64]: lane l, blk holds output bin 64*blk + pos[l]."""
    ar = np.empty((15, 64), f32)
    ai = np.empty((15, 64), f32)
    for blk in range(15):
        idx = 15 * lam(L) + c_of_blk(blk)
        ar[blk], ai[blk] = x_r[idx], x_i[idx]
    pos = np.zeros(64, int)
    ar, ai = radix4_stage(ar, ai, 0, pos, 0)
    pos = SIG[L & 3]
    ar, ai = radix4_stage(ar, ai, 2, pos, 60)
    pos = pos + 4 * SIG[(L >> 2) & 3]
    ar, ai = radix4_stage(ar, ai, 4, pos, 15)
    pos = pos + 16 * SIG[(L >> 4) & 3]
    # radix 3 (kiss_fft.c:201-225): blocks 3u, 3u+1, 3u+2; twiddles tw[5q], tw[10q], q = pos
    epi3i = TWI[5 * 64]
    half = f32(.5)
    for u in range(5):
        f0r, f0i = ar[3 * u], ai[3 * u]
        s1r, s1i = cmul(ar[3 * u + 1], ai[3 * u + 1], TWR[5 * pos], TWI[5 * pos])
        s2r, s2i = cmul(ar[3 * u + 2], ai[3 * u + 2], TWR[10 * pos], TWI[10 * pos])
        s3r, s3i = (s1r + s2r).astype(f32), (s1i + s2i).astype(f32)
        s0r, s0i = (s1r - s2r).astype(f32), (s1i - s2i).astype(f32)
        fmr = (f0r - (s3r * half).astype(f32)).astype(f32)
        fmi = (f0i - (s3i * half).astype(f32)).astype(f32)
        s0r = (s0r * epi3i).astype(f32)
        s0i = (s0i * epi3i).astype(f32)
        ar[3 * u], ai[3 * u] = (f0r + s3r).astype(f32), (f0i + s3i).astype(f32)
        ar[3 * u + 2], ai[3 * u + 2] = (fmr + s0i).astype(f32), (fmi - s0r).astype(f32)
        ar[3 * u + 1], ai[3 * u + 1] = (fmr - s0i).astype(f32), (fmi + s0r).astype(f32)
    # radix 5 (kiss_fft.c:269-302): blocks t, t+3, t+6, t+9, t+12; j = 64*t + pos
    yar, yai, ybr, ybi = TWR[192], TWI[192], TWR[384], TWI[384]
    for t in range(3):
        j = 64 * t + pos
        s0r, s0i = ar[t].copy(), ai[t].copy()
        s1r, s1i = cmul(ar[t + 3], ai[t + 3], TWR[j], TWI[j])
        s2r, s2i = cmul(ar[t + 6], ai[t + 6], TWR[2 * j], TWI[2 * j])
        s3r, s3i = cmul(ar[t + 9], ai[t + 9], TWR[3 * j], TWI[3 * j])
        s4r, s4i = cmul(ar[t + 12], ai[t + 12], TWR[(4 * j) % 960], TWI[(4 * j) % 960])
        s7r, s7i = (s1r + s4r).astype(f32), (s1i + s4i).astype(f32)
        s10r, s10i = (s1r - s4r).astype(f32), (s1i - s4i).astype(f32)
        s8r, s8i = (s2r + s3r).astype(f32), (s2i + s3i).astype(f32)
        s9r, s9i = (s2r - s3r).astype(f32), (s2i - s3i).astype(f32)
        ar[t] = (s0r + (s7r + s8r).astype(f32)).astype(f32)
        ai[t] = (s0i + (s7i + s8i).astype(f32)).astype(f32)
        s5r = (s0r + ((s7r * yar).astype(f32) + (s8r * ybr).astype(f32)).astype(f32)).astype(f32)
        s5i = (s0i + ((s7i * yar).astype(f32) + (s8i * ybr).astype(f32)).astype(f32)).astype(f32)
        s6r = ((s10i * yai).astype(f32) + (s9i * ybi).astype(f32)).astype(f32)
        s6i = (-((s10r * yai).astype(f32) + (s9r * ybi).astype(f32)).astype(f32)).astype(f32)
        ar[t + 3], ai[t + 3] = (s5r - s6r).astype(f32), (s5i - s6i).astype(f32)
        ar[t + 12], ai[t + 12] = (s5r + s6r).astype(f32), (s5i + s6i).astype(f32)
        s11r = (s0r + ((s7r * ybr).astype(f32) + (s8r * yar).astype(f32)).astype(f32)).astype(f32)
        s11i = (s0i + ((s7i * ybr).astype(f32) + (s8i * yar).astype(f32)).astype(f32)).astype(f32)
        s12r = ((s9i * yai).astype(f32) - (s10i * ybi).astype(f32)).astype(f32)
        s12i = ((s10r * ybi).astype(f32) - (s9r * yai).astype(f32)).astype(f32)
        ar[t + 6], ai[t + 6] = (s11r + s12r).astype(f32), (s11i + s12i).astype(f32)
        ar[t + 9], ai[t + 9] = (s11r - s12r).astype(f32), (s11i - s12i).astype(f32)
    return ar, ai, pos


def main():
    from oracle.binding import Oracle
    rng = np.random.Generator(np.random.PCG64(3))
    bad = 0
    for trial in range(40):
        if trial == 0:
            x = np.zeros((960, 2), f32)
        elif trial == 1:
            x = np.zeros((960, 2), f32); x[5, 0] = 1000; x[77, 1] = -3
        elif trial < 20:  # real input (imaginary part +0): the analysis windows
            x = np.zeros((960, 2), f32); x[:, 0] = rng.normal(0, 3000, 960).astype(f32)
        else:             # Hermitian-like complex input: the synthesis side
            x = rng.normal(0, 10 ** rng.uniform(-3, 3), (960, 2)).astype(f32)
            if trial % 3 == 0:
                x[rng.integers(0, 960, 300)] = 0
                x[rng.integers(0, 960, 30), 1] = -0.0
        want = Oracle.fft(x.reshape(-1).copy()).reshape(960, 2)
        scale = f32(1.0) / f32(960)  # T_FFT_SCALE
        xr, xi = (scale * x[:, 0]).astype(f32), (scale * x[:, 1]).astype(f32)
        yr, yi, pos = regfft(xr, xi)
        got = np.empty((960, 2), f32)
        for blk in range(15):
            got[64 * blk + pos, 0] = yr[blk]
            got[64 * blk + pos, 1] = yi[blk]
        ne = got.view(np.uint32) != want.view(np.uint32)
        if ne.any():
            bad += 1
            print(f"trial {trial}: {int(ne.sum())} words differ, first {np.argwhere(ne)[:4].tolist()}")
    print("pos (lane -> bin offset):", pos.tolist())
    print("OK: bit-identical to the oracle FFT on 40 inputs" if not bad else f"FAILED on {bad} inputs")
    return bad


if __name__ == "__main__":
    sys.exit(main())
