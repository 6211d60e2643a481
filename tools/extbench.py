#!/usr/bin/env python3
"""A/B of the Ext (fft_ext / ifft_ext, src/math/domain.rs:129-151) device path: round 3's form (ext_split_kernel -> batch of 4 base
transforms -> ext_join_kernel, one vector per call) against round 4's interleaved passes (no de-interleave, batched).
  python tools/extbench.py [old_lib.so]        default old lib: build/libtoyni_hip_r3.so (round 3's sources, built by hand)
Both libraries are driven through plain ctypes (the old one lacks the new entry points); the results of the two are compared
bit for bit before anything is timed.  Rates: Ext elements per second, and GB/s of the algorithmic 32 B per Ext element."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = 2013265921
old_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "build", "libtoyni_hip_r3.so")
new_path = os.path.join(ROOT, "toyni_amd", "lib", "libtoyni_hip.so")
dev = torch.device("cuda", 0)
torch.cuda.init()
stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def load(path):
    lib = ctypes.CDLL(path)
    lib.toyni_ntt_ctx_create.argtypes = [ctypes.c_uint32, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]
    lib.toyni_ntt_ext_device.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_int, ctypes.c_void_p]
    lib.toyni_ntt_device.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
    if hasattr(lib, "toyni_ntt_ext_batch_device"):
        lib.toyni_ntt_ext_batch_device.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint32, ctypes.c_int, ctypes.c_void_p]
    return lib


def ctx(lib, n):
    h = ctypes.c_void_p()
    assert lib.toyni_ntt_ctx_create(n, 0, ctypes.byref(h)) == 0
    return h


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


old, new = (load(old_path) if os.path.exists(old_path) else None), load(new_path)
print(f"old: {old_path if old else '(absent)'}\nnew: {new_path}")
for log_n, vecs in ((20, 64), (20, 1), (16, 1024), (24, 4), (12, 16384)):
    n = 1 << log_n
    x = torch.randint(0, P, (vecs * n * 4,), dtype=torch.int32, device=dev)
    cn = ctx(new, n)
    y_new = x.clone()
    assert new.toyni_ntt_ext_batch_device(cn, y_new.data_ptr(), y_new.data_ptr(), vecs, 7, 0, stream) == 0
    row = []
    if old:
        co = ctx(old, n)
        y_old = x.clone()
        for v in range(vecs):
            assert old.toyni_ntt_ext_device(co, y_old.data_ptr() + 16 * n * v, 7, 0, stream) == 0
        torch.cuda.synchronize()
        assert torch.equal(y_old, y_new), "round 3's and round 4's Ext transforms differ"
        del y_old
    buf = x.clone()
    p = buf.data_ptr()
    for rep in range(2):   # alternating
        if old:
            t_old = timed(lambda: [old.toyni_ntt_ext_device(co, p + 16 * n * v, 7, 0, stream) for v in range(vecs)], 5)
            row.append(("split/join, one vector per call (r3)", t_old))
        t_new1 = timed(lambda: [new.toyni_ntt_ext_device(cn, p + 16 * n * v, 7, 0, stream) for v in range(vecs)], 5)
        row.append(("interleaved, one vector per call", t_new1))
        t_newb = timed(lambda: new.toyni_ntt_ext_batch_device(cn, p, p, vecs, 7, 0, stream), 10)
        row.append(("interleaved, one batched call", t_newb))
        t_base = timed(lambda: new.toyni_ntt_device(cn, p, p, 4 * vecs, 0, stream), 10)
        row.append(("4x as many base transforms (same bytes)", t_base))
    print(f"n=2^{log_n} x {vecs} Ext vectors (coset forward, {vecs * n * 16 / 2**20:.0f} MiB):")
    for name, t in row:
        print(f"   {name:44s} {t * 1e3:9.3f} ms  {vecs * n / t / 1e9:7.2f} G Ext el/s  {32.0 * vecs * n / t / 1e9:7.0f} GB/s algorithmic")
    del x, y_new, buf
