#!/usr/bin/env python3
"""Alternating A/B of two BUILDS of the library on one box (has a change outside a kernel's hot path moved its speed?):
  python tools/ab_lib.py build/libtoyni_r04.so toyni_amd/lib/libtoyni_hip.so [log_n:batch ...]
One child process per (library, repetition); only round-1 entry points are used (toyni_ntt_ctx_create, toyni_ntt_device), so any
round's library loads.  Prints ms per forward + inverse (median of five 5-step windows) for A, B, A, B."""
import ctypes
import os
import subprocess
import sys


def child(lib_path, log_n, batch):
    import torch
    lib = ctypes.CDLL(lib_path)
    vp = ctypes.c_void_p
    lib.toyni_ntt_ctx_create.argtypes = [ctypes.c_uint32, ctypes.c_int, ctypes.POINTER(vp)]
    lib.toyni_ntt_device.argtypes = [vp, vp, vp, ctypes.c_size_t, ctypes.c_int, vp]
    h = vp()
    assert lib.toyni_ntt_ctx_create(1 << log_n, 0, ctypes.byref(h)) == 0
    dev = torch.device("cuda", 0)
    data = torch.randint(0, 2013265921, (batch << log_n,), dtype=torch.int32, device=dev)
    s = torch.cuda.current_stream().cuda_stream

    def step():
        assert lib.toyni_ntt_device(h, data.data_ptr(), data.data_ptr(), batch, 0, s or None) == 0
        assert lib.toyni_ntt_device(h, data.data_ptr(), data.data_ptr(), batch, 1, s or None) == 0

    for _ in range(5):
        step()
    torch.cuda.synchronize()
    ws = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5):
            step()
        b.record()
        torch.cuda.synchronize()
        ws.append(a.elapsed_time(b) / 5)
    print(f"{sorted(ws)[2]:.4f}")


if __name__ == "__main__":
    if sys.argv[1] == "child":
        child(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]))
        sys.exit(0)
    libs = [os.path.abspath(p) for p in sys.argv[1:3]]
    specs = sys.argv[3:] or ["20:1024"]
    for spec in specs:
        log_n, batch = spec.split(":")
        row = []
        for rep in range(3):
            for p in libs:
                out = subprocess.run([sys.executable, os.path.abspath(__file__), "child", p, log_n, batch], capture_output=True, text=True, timeout=300)
                row.append((os.path.basename(p), out.stdout.strip().splitlines()[-1] if out.returncode == 0 and out.stdout.strip() else "FAILED " + out.stderr[-200:]))
        print(f"n=2^{log_n} x {batch}: " + "  ".join(f"{n}: {v} ms" for n, v in row), flush=True)
