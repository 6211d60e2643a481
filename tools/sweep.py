#!/usr/bin/env python3
"""Throughput of the device-resident forward transform for every size 2^4 .. 2^27 at about 1 GiB of data per launch
sequence (batch = 2^28 / n).  Prints one line per size: passes, ms, elements/s, actual GB/s moved (8 B/element/pass).
SWEEP_EXT=1: the same data as Ext vectors through the interleaved passes.  SWEEP_INVERSE=1, SWEEP_SHIFT=<coset shift>, SWEEP_LDE=<log2
blow-up> (output elements are counted): the other forms of the same entry points.
With the measurement build (TOYNI_LIB_OVERRIDE=toyni_amd/lib/libtoyni_hip_tools.so) and SWEEP_PASSES=1: each pass kernel's time."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import toyni_amd  # noqa: E402

P = 2013265921
TOTAL = 1 << 28


def main():
    dev = torch.device("cuda", 0)
    data = torch.randint(0, P, (TOTAL,), dtype=torch.int32, device=dev)
    ptr = data.data_ptr()
    out = torch.empty(TOTAL, dtype=torch.int32, device=dev) if os.environ.get("SWEEP_LDE") else None
    stream = torch.cuda.current_stream().cuda_stream
    lo, hi = (int(v) for v in os.environ.get("SWEEP_RANGE", "4:28").split(":"))
    for log_n in range(lo, hi):
        n = 1 << log_n
        batch = int(os.environ.get("SWEEP_BATCH", TOTAL // n))   # SWEEP_BATCH=1: latency of one transform
        if batch * n > TOTAL:
            sys.exit(f"SWEEP_BATCH={batch} x n=2^{log_n} exceeds the {TOTAL}-element buffer of this tool")
        ctx = toyni_amd.NttContext(n)
        inverse = bool(os.environ.get("SWEEP_INVERSE"))
        shift = int(os.environ.get("SWEEP_SHIFT", "1"))          # coset shift (1 = plain)
        lde = int(os.environ.get("SWEEP_LDE", "0"))              # > 0: low-degree extension by 2^lde (out of place, forward, into `out`)
        if lde:
            if lde > log_n or (os.environ.get("SWEEP_EXT") and batch < 4):
                continue
            if os.environ.get("SWEEP_EXT"):
                f = lambda: ctx.lde_ext_device(ptr, out.data_ptr(), batch // 4, lde, shift, stream=stream)  # noqa: E731
            else:
                f = lambda: ctx.lde_device(ptr, out.data_ptr(), batch, lde, shift, stream=stream)  # noqa: E731
        elif os.environ.get("SWEEP_EXT"):    # the same bytes as Ext vectors (AoS, four interleaved coordinates): batch / 4 vectors
            if batch < 4:
                continue
            f = lambda: ctx.run_device_ext_batch(ptr, ptr, batch // 4, inverse, shift=shift, stream=stream)  # noqa: E731
        else:
            f = lambda: ctx.run_device(ptr, ptr, batch, inverse, stream=stream, shift=shift)  # noqa: E731
        for _ in range(3 if log_n > lo else 60):   # the first size also brings the chip out of its idle clocks
            f()
        torch.cuda.synchronize()
        reps = 10 if batch > 1 else 200
        windows = []
        for _ in range(5):   # five windows of `reps` calls: the median (single windows differ by up to 15 % from box to box at 2^18 / 2^19)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(reps):
                f()
            b.record()
            torch.cuda.synchronize()
            windows.append(a.elapsed_time(b) / reps)
        ms = sorted(windows)[2]
        spread = (max(windows) - min(windows)) / ms
        # one launch per call where the library says so (toyni_ntt_ctx_passes_for: the one-wave-per-transform kernel of 2^11, the single-sweep
        # LDS kernel of large batches of 2^12 / 2^13, the two-pass plan of 2^21); LDE / Ext forms: the plain plan's count
        sweeps = ctx.passes if (lde or os.environ.get("SWEEP_EXT")) else ctx.passes_for(batch)
        print(f"n=2^{log_n:<2d} batch={batch:<9d} sweeps={sweeps} {ms:8.4f} ms  {batch * n / ms / 1e6:8.1f} Gelem/s  "
              f"{8.0 * sweeps * batch * n / ms / 1e9:7.2f} TB/s moved  (windows +-{50 * spread:.1f} %)", flush=True)
        if toyni_amd._lib.HAS_TOOLS and os.environ.get("SWEEP_PASSES"):   # TOYNI_LIB_OVERRIDE=.../libtoyni_hip_tools.so
            per = ctx.profile_passes(ptr, batch, False, reps=5, stream=stream)
            print("      per pass: " + "  ".join(f"{t:.4f} ms ({8.0 * batch * n / t / 1e9:.2f} TB/s)" for t in per), flush=True)
        ctx.destroy()


if __name__ == "__main__":
    main()
