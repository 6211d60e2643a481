#!/usr/bin/env python3
"""Small, torch-free driver for rocprofv3: a few launches of every hot kernel on synthetic data.

  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python3 tools/prof_ntt.py
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 tools/prof_ntt.py
"""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("TOYNI_HIP_RUNTIME", "system")  # torch-free driver: the ROCm install's own runtime
import toyni_amd  # noqa: E402
from toyni_amd._lib import check, lib  # noqa: E402

P = 2013265921
LOG_N = int(os.environ.get("PROF_LOG_N", "20"))
BATCH = int(os.environ.get("PROF_BATCH", "256"))
REPS = int(os.environ.get("PROF_REPS", "5"))


def dmalloc(nbytes):
    p = ctypes.c_void_p()
    check(lib.toyni_malloc(ctypes.byref(p), nbytes), "malloc")
    return p.value


def main():
    n = 1 << LOG_N
    rng = np.random.default_rng(7)
    block = rng.integers(0, P, size=min(BATCH, 16) * n, dtype=np.uint32)
    d = dmalloc(BATCH * n * 4)
    for off in range(0, BATCH * n * 4, block.nbytes):
        lib.toyni_memcpy_h2d(d + off, block.ctypes.data, min(block.nbytes, BATCH * n * 4 - off))
    ctx = toyni_amd.NttContext(n)
    for _ in range(REPS):
        ctx.run_device(d, d, BATCH, False)
        ctx.run_device(d, d, BATCH, True)
    ctx.synchronize()
    back = np.empty(n, dtype=np.uint32)
    lib.toyni_memcpy_d2h(back.ctypes.data, d, back.nbytes)
    assert (back == block[:n]).all(), "round trip changed the data"

    if os.environ.get("PROF_EXTRAS", "1") == "1":
        # single 2^24 transform (3 passes) and the FRI fold on a 2^27 layer (512 MiB in, beyond the Infinity Cache)
        c24 = toyni_amd.NttContext(1 << 24)
        for _ in range(REPS):
            c24.run_device(d, d, 1, False)
        c27 = toyni_amd.NttContext(1 << 27)
        o = dmalloc((1 << 26) * 4)
        for _ in range(REPS):
            toyni_amd.fri_fold_device(c27, d, o, 1 << 27, 123456789, 7)
        c27.synchronize()
    print("prof driver done")


if __name__ == "__main__":
    main()
