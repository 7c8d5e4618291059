#!/usr/bin/env python3
"""What the host link gives: pinned host <-> HBM copies of the size the host-fed path moves (hipMemcpyAsync through torch),
one direction at a time and both at once on two streams.  The yardstick for `bench.py --host-io [--s16]`.
usage: tools/pcie_peak.py [MiB per copy = 252]"""
import sys
import time

import torch

mb = int(sys.argv[1]) if len(sys.argv) > 1 else 252
n = mb << 20
h_a = torch.empty(n, dtype=torch.uint8).pin_memory()
h_b = torch.empty(n, dtype=torch.uint8).pin_memory()
d_a = torch.empty(n, dtype=torch.uint8, device="cuda")
d_b = torch.empty(n, dtype=torch.uint8, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def timed(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def h2d():
    with torch.cuda.stream(s1):
        d_a.copy_(h_a, non_blocking=True)


def d2h():
    with torch.cuda.stream(s2):
        h_b.copy_(d_b, non_blocking=True)


def both():
    h2d()
    d2h()


t_up, t_dn, t_bi = timed(h2d), timed(d2h), timed(both)
print(f"pinned <-> HBM, {mb} MiB per copy: H2D {n / t_up / 1e9:.1f} GB/s, D2H {n / t_dn / 1e9:.1f} GB/s, "
      f"both at once {n / t_bi / 1e9:.1f} GB/s per direction ({2 * n / t_bi / 1e9:.1f} GB/s total)")
