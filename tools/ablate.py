#!/usr/bin/env python3
"""Timing-only ablation of the pass kernels (diagnostic): the measurement build compiled with -DTOYNI_ABLATE=0|1|2|3 (bit 0: tile
loads replaced by register arithmetic, bit 1: tile stores never taken) prices the VALU / LDS side of every pass against its HBM side.
Results of the ablated builds are garbage by construction; only the times mean anything.

  python tools/ablate.py build            # here (hipcc cross-compiles): build/libtoyni_ablate{0,1,2,3}.so
  python tools/ablate.py run [log_n batch]...   # on the GPU box: per-pass forward times for every build
"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIBS = {a: os.path.join(ROOT, "build", f"libtoyni_ablate{a}.so") for a in (0, 1, 2, 3)}


def build():
    import __graft_entry__ as entry
    os.makedirs(os.path.join(ROOT, "build"), exist_ok=True)
    procs = []
    for a, out in LIBS.items():
        cmd = [entry._hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-DTOYNI_TOOLS", f"-DTOYNI_ABLATE={a}",
               "-I", os.path.join(ROOT, "include"), "-o", out, os.path.join(ROOT, "toyni_amd", "csrc", "toyni_hip.hip")]
        procs.append(subprocess.Popen(cmd, cwd=ROOT))
    assert all(p.wait() == 0 for p in procs)


def child(lib_path, cases):
    lib = ctypes.CDLL(lib_path)
    vp, ci, sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t
    lib.toyni_ntt_ctx_create.argtypes = [ctypes.c_uint32, ci, ctypes.POINTER(vp)]
    lib.toyni_malloc.argtypes = [ctypes.POINTER(vp), sz]
    lib.toyni_ntt_profile_passes.argtypes = [vp, vp, sz, ci, ci, ctypes.POINTER(ctypes.c_float), vp]
    lib.toyni_ntt_ctx_passes_for.argtypes = [vp, sz]
    lib.toyni_ntt_ctx_destroy.argtypes = [vp]
    lib.toyni_free.argtypes = [vp]
    for log_n, batch in cases:
        h, d = vp(), vp()
        assert lib.toyni_ntt_ctx_create(1 << log_n, -1, ctypes.byref(h)) == 0
        assert lib.toyni_malloc(ctypes.byref(d), batch << (log_n + 2)) == 0
        ms = (ctypes.c_float * 3)()
        best = None
        for _ in range(3):
            assert lib.toyni_ntt_profile_passes(h, d, batch, 0, 10, ms, None) == 0
            cur = [ms[p] for p in range(lib.toyni_ntt_ctx_passes_for(h, batch))]
            best = cur if best is None else [min(a, b) for a, b in zip(best, cur)]
        print(f"  2^{log_n} x {batch}: " + "  ".join(f"{t:.3f}" for t in best) + f"   sum {sum(best):.3f} ms", flush=True)
        lib.toyni_ntt_ctx_destroy(h)
        lib.toyni_free(d)


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build()
    elif sys.argv[1] == "child":
        child(sys.argv[2], [(int(a), int(b)) for a, b in zip(sys.argv[3::2], sys.argv[4::2])])
    else:
        cases = sys.argv[2:] or ["20", "1024", "24", "64", "16", "16384"]
        for a, path in LIBS.items():
            print(f"ablate={a} ({['full kernel', 'no tile loads', 'no tile stores', 'neither: VALU + LDS + barriers'][a]}), forward pass times in ms (min of 3 x 10 launches)", flush=True)
            subprocess.check_call([sys.executable, os.path.abspath(__file__), "child", path] + cases)
