"""Builds and runs tests/cpp/test_ntt_gpu.cpp: the reference's GPU tests (src/ntt.rs:253-311) in C++ over the
host mirror toyni_amd/csrc/host/toyni_ntt.hpp, linked against libtoyni_hip.so (the C ABI) and the oracle."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build():
    import __graft_entry__ as entry
    entry.build_hip()
    out = os.path.join(ROOT, "tests", "emu", "build", "test_ntt_gpu")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    obj = out + "_oracle.o"
    subprocess.check_call(["gcc", "-O2", "-c", "-o", obj, os.path.join(ROOT, "oracle", "toyni_oracle.c")])
    libdir = os.path.join(ROOT, "toyni_amd", "lib")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "include"), "-o", out, os.path.join(ROOT, "tests", "cpp", "test_ntt_gpu.cpp"), obj,
                           "-L", libdir, "-ltoyni_hip", f"-Wl,-rpath,{libdir}", "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"])
    return out


def test_cpp_host_mirror_builds_and_degrades_without_gpu():
    exe = _build()
    import toyni_amd
    if toyni_amd.gpu_available():
        pytest.skip("covered by the gpu-marked run")
    res = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0 and "GPU available: 0" in res.stdout and "CPP OK" in res.stdout  # src/ntt.rs:265-268 self-skip


@pytest.mark.gpu
def test_cpp_reference_gpu_tests():
    exe = _build()
    res = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    assert "GPU available: 1" in res.stdout and "CPP OK" in res.stdout
