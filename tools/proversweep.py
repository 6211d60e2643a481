#!/usr/bin/env python3
"""The Merkle commitment and the pointwise prover steps across sizes (device-resident): us per call and the rate of the unit each is
priced on -- looks for dips at the launchers' size gates (two-wave node hash for levels <= 2^14, single-workgroup tail <= 512)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import toyni_amd  # noqa: E402
from toyni_amd._lib import lib  # noqa: E402

P = 2013265921
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream


def timed(fn, reps):
    """Median of three windows; a call that turns out shorter than 50 us is re-timed with at least 100 calls per window (a 10-call window
    reads up to 1.5 us more -- and one such window once produced the N = 2^18 outlier of profiles/r04_proversweep.txt, see
    profiles/r05_proversweep_2p18.txt)."""
    def windows(k):
        out = []
        for _ in range(3):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(k):
                fn()
            b.record()
            torch.cuda.synchronize()
            out.append(a.elapsed_time(b) / k * 1e-3)
        return sorted(out)[1]
    fn()
    torch.cuda.synchronize()
    t = windows(reps)
    if t < 50e-6 and reps < 100:
        t = windows(100)
    return t


print("# Merkle commitment (salted leaves + all node levels): n leaves -> 3 n - 2 compressions (one per leaf, two per node)")
for lg in range(4, 25):
    n = 1 << lg
    vals = torch.randint(0, P, (n,), dtype=torch.int32, device=dev)
    salts = torch.randint(0, 255, (n, 16), dtype=torch.uint8, device=dev)
    levels = torch.empty((lib.toyni_merkle_total_digests(n), 32), dtype=torch.uint8, device=dev)
    reps = 100 if lg < 18 else 10
    t = timed(lambda: toyni_amd.merkle_commit_device(vals.data_ptr(), salts.data_ptr(), n, levels.data_ptr(), stream=stream), reps)
    print(f"n=2^{lg:<2d} {t * 1e6:9.1f} us  {(3 * n - 2) / t / 1e9:6.2f} G compressions/s", flush=True)
    del vals, salts, levels

print("# quotient / DEEP / domain points on an LDE coset of N points (blow-up 32)")
for lg in range(10, 25):
    N = 1 << lg
    ctx = toyni_amd.NttContext(N)
    t_lde = torch.randint(0, P, (N,), dtype=torch.int32, device=dev)
    q = torch.empty(N, dtype=torch.int32, device=dev)
    d = torch.empty(N, dtype=torch.int32, device=dev)
    reps = 100 if lg < 18 else 10
    tq = timed(lambda: toyni_amd.prover.fib_quotient_device(ctx, t_lde.data_ptr(), 0, q.data_ptr(), 5, 7, stream=stream), reps)
    td = timed(lambda: toyni_amd.prover.fib_deep_device(ctx, t_lde.data_ptr(), q.data_ptr(), d.data_ptr(), 5, 7, 123456, [1, 2, 3, 4], stream=stream), reps)
    tp = timed(lambda: ctx.domain_elements_device(d.data_ptr(), N, 7, stream=stream), reps)
    print(f"N=2^{lg:<2d} quotient {tq * 1e6:8.1f} us {8.0 * N / tq / 1e9:6.0f} GB/s | DEEP {td * 1e6:8.1f} us {12.0 * N / td / 1e9:6.0f} GB/s | points {tp * 1e6:8.1f} us {4.0 * N / tp / 1e9:6.0f} GB/s",
          flush=True)
    ctx.destroy()
