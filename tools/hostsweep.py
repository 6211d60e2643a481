#!/usr/bin/env python3
"""The reference-shaped host-slice entry points across sizes (PCIe inclusive, pageable numpy arrays = what a Rust Vec is):
ntt_cuda-shaped toyni_ntt_host, the coset form, BabyBearDomain::fft with a 32x shorter coefficient vector (toyni_lde_host), fft_ext,
and fri_fold on host slices.  us per call.  Outputs go to preallocated arrays: a fresh 32 MiB+ numpy array per call is fresh
mmap'ed pages per call, and their first-touch faults during the download (3 ms at 32 MiB) would be charged to the library."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import toyni_amd  # noqa: E402

P = 2013265921
rng = np.random.default_rng(5)


def wall(fn, reps):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps


for lg in range(3, 25):
    n = 1 << lg
    reps = 200 if lg < 16 else (20 if lg < 22 else 5)
    ctx = toyni_amd.ntt.get_or_create_ctx(n)
    v = rng.integers(0, P, n, dtype=np.uint64)
    t_plain = wall(lambda: ctx.run_host(v, False), reps)
    t_coset = wall(lambda: ctx.run_host(v, True, shift=7), reps)
    c = v[: max(1, n >> 5)].copy()
    out_n = np.empty(n, dtype=np.uint64)
    t_lde = wall(lambda: ctx.lde_host(c, shift=7, out=out_n), reps)
    line = f"n=2^{lg:<2d} ntt_host {t_plain * 1e6:9.1f} us | coset intt {t_coset * 1e6:9.1f} us | lde_host (n/32 coeffs) {t_lde * 1e6:9.1f} us"
    if lg <= 22:
        v4 = rng.integers(0, P, 4 * n, dtype=np.uint64)
        t_ext = wall(lambda: ctx.run_host_ext(v4, False, shift=7), reps)
        line += f" | ext_host {t_ext * 1e6:9.1f} us"
    if lg >= 1:
        xs = rng.integers(1, P, n // 2, dtype=np.uint64)
        out_h = np.empty(n // 2, dtype=np.uint64)
        t_fold = wall(lambda: toyni_amd.fri_fold(v, xs, 12345, out=out_h), reps)
        line += f" | fri_fold {t_fold * 1e6:9.1f} us"
    print(line, flush=True)
