#!/usr/bin/env python3
"""Fold throughput across layer sizes 2^10 .. 2^27 (device-resident): structured points (toyni_fri_fold_device), explicit points
(toyni_fri_fold_xs_device) and Ext values (toyni_fri_fold_ext_device; layers up to 2^25 Ext elements).  GB/s of the algorithmic bytes
(6 / 8 / 24 B per input element).  Looks for dips at the launcher's size gates."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import toyni_amd  # noqa: E402

P = 2013265921
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream
ctx = toyni_amd.NttContext(1 << 27)
big = torch.randint(0, P, (1 << 27,), dtype=torch.int32, device=dev)
xs = torch.randint(1, P, (1 << 26,), dtype=torch.int32, device=dev)
out = torch.empty(1 << 26, dtype=torch.int32, device=dev)


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e-3


for lg in range(10, 28):
    m = 1 << lg
    reps = 200 if lg < 23 else 20   # launch-bound sizes: 20-call windows scatter between 4 and 12 us
    t_s = timed(lambda: toyni_amd.fri_fold_device(ctx, big.data_ptr(), out.data_ptr(), m, 123456789, 7, stream=stream), reps)
    t_x = timed(lambda: toyni_amd.fri_fold_xs_device(big.data_ptr(), xs.data_ptr(), out.data_ptr(), m, 123456789, stream=stream), reps)
    line = f"m=2^{lg:<2d} structured {t_s * 1e6:9.1f} us {6.0 * m / t_s / 1e9:7.0f} GB/s | explicit points {t_x * 1e6:9.1f} us {8.0 * m / t_x / 1e9:7.0f} GB/s"
    if lg <= 25:
        t_e = timed(lambda: toyni_amd.fri_fold_ext_device(ctx, big.data_ptr(), out.data_ptr(), m, [11, 22, 33, 44], 7, stream=stream), reps)
        line += f" | Ext {t_e * 1e6:9.1f} us {24.0 * m / t_e / 1e9:7.0f} GB/s"
    print(line, flush=True)
