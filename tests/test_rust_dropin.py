"""The Rust drop-in, as far as it can be verified without rustc: following INTEGRATION.md section 1 literally
must produce a build.  The test lays out a scratch crate, copies the files INTEGRATION.md lists into hip/, runs the EXACT
hipcc and ar commands that rust/build.rs would run (parsed out of build.rs, so the two cannot drift), and links a host program
against the resulting static archive with exactly the libraries build.rs tells cargo to link (reference: build.rs:75-116)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _integration_file_list():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    row = next(l for l in text.splitlines() if l.startswith("| `cuda/ntt_kernel.cu`"))
    m = re.search(r"`toyni_amd/csrc/\{([^}]*)\}`", row)
    assert m and "`include/toyni_hip.h`" in row and "copy to `hip/`" in row, row
    return [os.path.join(ROOT, "toyni_amd", "csrc", f) for f in m.group(1).split(",")] + [os.path.join(ROOT, "include", "toyni_hip.h")]


def _build_rs_commands():
    text = open(os.path.join(ROOT, "rust", "build.rs")).read()
    m = re.search(r"Command::new\(&hipcc\)\s*\.args\(\[([^\]]*)\]\)\s*\.arg\(&obj\)", text)
    assert m, "hipcc invocation not found in rust/build.rs"
    hipcc_args = re.findall(r'"([^"]*)"', m.group(1))
    assert re.search(r'Command::new\("ar"\)\.arg\("rcs"\)\.arg\(&lib\)\.arg\(&obj\)', text), "ar invocation not found"
    libs = re.findall(r'cargo:rustc-link-lib=(\w+)=(\w[\w+]*)', text)
    rerun = re.findall(r"cargo:rerun-if-changed=hip/([\w.]+)", text)
    return hipcc_args, libs, rerun


def test_build_rs_tracks_every_copied_file():
    _, _, rerun = _build_rs_commands()
    listed = {os.path.basename(f) for f in _integration_file_list()}
    for f in listed:
        assert f in rerun, f"rust/build.rs has no rerun-if-changed line for hip/{f}"
    # and INTEGRATION.md lists every source of the translation unit (csrc/host/ is the C++ mirror, not part of the crate)
    csrc = os.path.join(ROOT, "toyni_amd", "csrc")
    for f in os.listdir(csrc):
        if f.endswith((".hip", ".hpp", ".h")):
            assert f in listed, f"toyni_amd/csrc/{f} is missing from INTEGRATION.md section 1 (the crate's hip/ directory would not build)"


def test_following_integration_md_produces_a_linkable_archive(tmp_path):
    import __graft_entry__ as entry
    crate = tmp_path / "toyni"
    (crate / "hip").mkdir(parents=True)
    for f in _integration_file_list():
        assert os.path.exists(f), f"INTEGRATION.md lists {f}, which does not exist"
        shutil.copy(f, crate / "hip")
    # every csrc file the translation unit includes must be on the list (a file missing from INTEGRATION.md breaks the crate)
    out = tmp_path / "out"
    out.mkdir()
    hipcc_args, libs, _ = _build_rs_commands()
    assert hipcc_args[-1] == "-o" and "--offload-arch=gfx950" in hipcc_args and sum(a.startswith("--offload-arch") for a in hipcc_args) == 1
    obj, lib = out / "toyni_hip.o", out / "libtoyni_hip.a"
    res = subprocess.run([entry._hipcc()] + hipcc_args + [str(obj)], cwd=crate, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    subprocess.check_call(["ar", "rcs", str(lib), str(obj)], cwd=crate)
    # what cargo would do with build.rs's output: link the archive statically, then the dylibs, in that order
    assert libs == [("static", "toyni_hip"), ("dylib", "amdhip64"), ("dylib", "stdc++")], libs
    host_obj = tmp_path / "host.o"
    oracle_obj = tmp_path / "oracle.o"
    subprocess.check_call(["gcc", "-O2", "-c", "-o", str(oracle_obj), os.path.join(ROOT, "oracle", "toyni_oracle.c")])
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", str(crate / "hip"), "-c", "-o", str(host_obj),
                           os.path.join(ROOT, "tests", "cpp", "test_ntt_gpu.cpp")])
    exe = tmp_path / "host"
    # linked with plain cc-style driver flags (no hipcc at link time: rustc links with cc)
    res = subprocess.run(["gcc", "-o", str(exe), str(host_obj), str(oracle_obj), f"-L{out}", "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib",
                          "-Wl,-Bstatic", "-ltoyni_hip", "-Wl,-Bdynamic", "-lamdhip64", "-lstdc++", "-lm", "-lpthread"],
                         capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-3000:]
    # all ten symbols of the reference's extern block (src/ntt.rs:95-110; cudaGetDeviceCount comes from libcudart there) are in the archive
    syms = subprocess.run(["nm", "--defined-only", str(lib)], capture_output=True, text=True).stdout
    for name in ("ntt_ctx_create", "ntt_ctx_destroy", "ntt_run_inplace", "intt_run_inplace", "cuda_malloc", "cuda_free",
                 "cuda_copy_to_device", "cuda_copy_from_device", "cuda_get_error_string", "cudaGetDeviceCount", "toyni_device_count"):
        assert re.search(rf"\bT {name}\b", syms), f"{name} not defined in libtoyni_hip.a"
    run = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0 and "CPP OK" in run.stdout, run.stdout[-2000:] + run.stderr[-2000:]
    import toyni_amd
    assert ("GPU available: 1" if toyni_amd.gpu_available() else "GPU available: 0") in run.stdout


@pytest.mark.gpu
def test_static_archive_runs_the_reference_gpu_tests(tmp_path):
    """The same on the GPU box: the statically linked host program runs the reference's three GPU tests (src/ntt.rs:253-311)."""
    test_following_integration_md_produces_a_linkable_archive(tmp_path)
