#!/usr/bin/env python3
"""Latency of single (or few) transforms, device-resident, kernel-only: us per call for n = 2^LO .. 2^HI at batch 1, 2, 4, 8, 16.
The three-step latency shapes are chosen per launch by tile count (TOYNI_P3_TILES = N: launches of <= 2^N 32-wide tiles;
-1 = never): run this tool under different values to find the crossover.  LAT_RANGE=16:25 LAT_BATCHES=1,2,4"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import toyni_amd  # noqa: E402

P = 2013265921


def time_us(fn, reps):
    fn(); fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def main():
    dev = torch.device("cuda", 0)
    lo, hi = (int(v) for v in os.environ.get("LAT_RANGE", "14:25").split(":"))
    batches = [int(v) for v in os.environ.get("LAT_BATCHES", "1,2,4,8,16").split(",")]
    stream = torch.cuda.current_stream().cuda_stream
    print(f"# TOYNI_P3_TILES={os.environ.get('TOYNI_P3_TILES', '(default)')}  us per call: forward / inverse / coset forward (shift 7)")
    for log_n in range(lo, hi):
        n = 1 << log_n
        ctx = toyni_amd.NttContext(n)
        for batch in batches:
            if batch * n > (1 << 28):
                continue
            data = torch.randint(0, P, (batch * n,), dtype=torch.int32, device=dev)
            p = data.data_ptr()
            reps = 200 if batch * n <= (1 << 22) else 30
            f = time_us(lambda: ctx.run_device(p, p, batch, False, stream=stream), reps)
            i = time_us(lambda: ctx.run_device(p, p, batch, True, stream=stream), reps)
            c = time_us(lambda: ctx.run_device(p, p, batch, False, stream=stream, shift=7), reps)
            print(f"n=2^{log_n:<2d} batch={batch:<3d} passes={ctx.passes}  {f:9.2f} {i:9.2f} {c:9.2f} us   {batch * n / f / 1e3:8.1f} Gelem/s forward", flush=True)
            del data
        ctx.destroy()


if __name__ == "__main__":
    main()
