#!/usr/bin/env python3
"""the N = 2^18 row of profiles/r04_proversweep.txt (quotient / DEEP / points 2.5-3x slower than both neighbours).
Repeats N = 2^17, 2^18, 2^19 three times each with 10 and with 100 calls per timed window, fresh context per repeat as the sweep does,
and once more with the context and buffers of the PREVIOUS size still alive (the sweep frees them before the next size)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import toyni_amd  # noqa: E402

P = 2013265921
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def row(lg, reps, keep=None):
    N = 1 << lg
    ctx = toyni_amd.NttContext(N)
    t_lde = torch.randint(0, P, (N,), dtype=torch.int32, device=dev)
    q = torch.empty(N, dtype=torch.int32, device=dev)
    d = torch.empty(N, dtype=torch.int32, device=dev)
    tq = timed(lambda: toyni_amd.prover.fib_quotient_device(ctx, t_lde.data_ptr(), 0, q.data_ptr(), 5, 7, stream=stream), reps)
    td = timed(lambda: toyni_amd.prover.fib_deep_device(ctx, t_lde.data_ptr(), q.data_ptr(), d.data_ptr(), 5, 7, 123456, [1, 2, 3, 4], stream=stream), reps)
    tp = timed(lambda: ctx.domain_elements_device(d.data_ptr(), N, 7, stream=stream), reps)
    print(f"N=2^{lg} reps={reps:<3d} quotient {tq:6.1f} us | DEEP {td:6.1f} us | points {tp:6.1f} us", flush=True)
    if keep is not None:
        keep.append((ctx, t_lde, q, d))
    else:
        ctx.destroy()


for rep in range(3):
    print(f"# repeat {rep}: the sweep's order (10 .. 17 at 100 calls per window first, as tools/proversweep.py does)")
    for lg in range(10, 18):
        row(lg, 100)
    for lg in (18, 19):
        row(lg, 10)
    for lg in (17, 18, 19):
        row(lg, 100)
    for lg in (17, 18, 19):
        row(lg, 10)
print("# buffers of every earlier size kept alive")
keep = []
for lg in (16, 17, 18, 19):
    row(lg, 10, keep)
for lg in (16, 17, 18, 19):
    row(lg, 100, keep)
