#!/usr/bin/env python3
"""Quick parity + timing check of one size through the device-resident entry points (used while bringing up new pass shapes):
forward against the oracle on sampled transforms, the whole batch through inverse(forward(x)) == x, a coset round trip, and the LDE by
2^5.  S3_LOGS="21 22" S3_BATCHES="1 3 4 16 128"."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle  # noqa: E402  (the checker)
import toyni_amd  # noqa: E402

P = 2013265921


def timed(f, reps=10):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    ws = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            f()
        b.record()
        torch.cuda.synchronize()
        ws.append(a.elapsed_time(b) / reps)
    return sorted(ws)[2]


def main():
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream().cuda_stream
    rng = np.random.default_rng(12345)
    for log_n in (int(v) for v in os.environ.get("S3_LOGS", "21").split()):
        n = 1 << log_n
        ctx = toyni_amd.NttContext(n)
        for batch in (int(v) for v in os.environ.get("S3_BATCHES", "1 3 4 16 128").split()):
            x = rng.integers(0, P, size=n * batch, dtype=np.uint32)
            d = torch.from_numpy(x.view(np.int32)).to(dev)
            o = torch.empty_like(d)
            for shift in (1, 7):
                ctx.run_device(d.data_ptr(), o.data_ptr(), batch, False, stream=stream, shift=shift)
                torch.cuda.synchronize()
                y = o.cpu().numpy().view(np.uint32)
                for t in sorted({0, batch // 2, batch - 1}):
                    want = oracle.domain_fft(x[t * n:(t + 1) * n].astype(np.uint64), n, shift)
                    assert (y[t * n:(t + 1) * n] == want).all(), f"2^{log_n} x{batch} shift {shift}: transform {t} differs"
                ctx.run_device(o.data_ptr(), o.data_ptr(), batch, True, stream=stream, shift=shift)
                torch.cuda.synchronize()
                assert (o.cpu().numpy().view(np.uint32) == x).all(), f"2^{log_n} x{batch} shift {shift}: round trip"
            ms = timed(lambda: ctx.run_device(d.data_ptr(), d.data_ptr(), batch, False, stream=stream), 10 if batch > 4 else 100)
            msi = timed(lambda: ctx.run_device(d.data_ptr(), d.data_ptr(), batch, True, stream=stream, shift=7), 10 if batch > 4 else 100)
            print(f"2^{log_n} x{batch}: parity ok; forward {ms:.4f} ms = {batch * n / ms / 1e6:.1f} Gel/s; inverse coset {msi:.4f} ms", flush=True)
        for batch in (1, 4, 64):
            z = 5
            c = rng.integers(0, P, size=(n >> z) * batch, dtype=np.uint32)
            dc = torch.from_numpy(c.view(np.int32)).to(dev)
            o = torch.empty(n * batch, dtype=torch.int32, device=dev)
            ctx.lde_device(dc.data_ptr(), o.data_ptr(), batch, z, 7, stream=stream)
            torch.cuda.synchronize()
            y = o.cpu().numpy().view(np.uint32)
            for t in sorted({0, batch - 1}):
                want = oracle.domain_fft(c[t * (n >> z):(t + 1) * (n >> z)].astype(np.uint64), n, 7)
                assert (y[t * n:(t + 1) * n] == want).all(), f"lde 2^{log_n} x{batch}: vector {t} differs"
            ms = timed(lambda: ctx.lde_device(dc.data_ptr(), o.data_ptr(), batch, z, 7, stream=stream), 10 if batch > 4 else 100)
            print(f"lde 2^{log_n - z} -> 2^{log_n} x{batch}: parity ok; {ms:.4f} ms", flush=True)
        ctx.destroy()
    print("S3CHECK OK", flush=True)


if __name__ == "__main__":
    t0 = time.time()
    main()
    print(f"({time.time() - t0:.0f} s)")
