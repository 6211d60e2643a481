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


REFERENCE = "/root/reference"   # present in the build container only (never on the GPU box)


def _load_apply():
    import importlib.util
    spec = importlib.util.spec_from_file_location("toyni_rust_apply", os.path.join(ROOT, "rust", "apply.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _patched_checkout(tmp_path):
    """A temporary copy of the reference checkout with rust/apply.py applied (build container only)."""
    dst = tmp_path / "toyni_checkout"
    shutil.copytree(REFERENCE, dst, ignore=shutil.ignore_patterns(".git", "target"))
    for d, _, names in os.walk(dst):
        os.chmod(d, 0o755)
        for n in names:
            os.chmod(os.path.join(d, n), 0o644)
    return dst, _load_apply().apply(str(dst))


def test_apply_script_lists_the_files_of_integration_md():
    assert sorted(_load_apply().HIP_SOURCES + ["toyni_hip.h"]) == sorted(os.path.basename(f) for f in _integration_file_list())


@pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="needs the reference checkout (build container only)")
def test_apply_script_patches_a_reference_checkout_mechanically(tmp_path):
    """the drop-in as a program.  rust/apply.py on a copy of the reference: the CPU half of src/ntt.rs is untouched, the
    `mod cuda` block is gone, exactly one real and one stub module took its place, the reference's public names are exported, the
    call sites of src/math/domain.rs still resolve, the GPU tests nested in the old block were carried over, the feature table and
    the build script are the new ones, cuda/ is gone and hip/ holds the library."""
    dst, report = _patched_checkout(tmp_path)
    ref_ntt = open(os.path.join(REFERENCE, "src", "ntt.rs")).read()
    new_ntt = open(dst / "src" / "ntt.rs").read()
    a, b = report["ntt_rs_block_lines"]
    ref_lines, new_lines = ref_ntt.splitlines(keepends=True), new_ntt.splitlines(keepends=True)
    assert 80 <= a <= 90 and 300 <= b <= 330, (a, b)                      # the GPU wrapper is src/ntt.rs:83-315
    assert ref_lines[:a - 1] == new_lines[:a - 1], "the CPU transform above the GPU block must stay byte for byte"
    assert "".join(ref_lines[b:]) == new_ntt[-len("".join(ref_lines[b:])):], "the CPU tests below the GPU block must stay byte for byte"
    assert not re.search(r"^\s*mod cuda\b", new_ntt, flags=re.M) and 'name = "ntt_cuda"' not in new_ntt and "pub use cuda::" not in new_ntt
    assert len(re.findall(r"^mod gpu \{", new_ntt, flags=re.M)) == 1 and len(re.findall(r"^mod gpu_absent \{", new_ntt, flags=re.M)) == 1
    for cfg in ('#[cfg(all(feature = "hip", has_hip))]', '#[cfg(all(feature = "hip", not(has_hip)))]'):
        # the four public aliases of the reference (src/ntt.rs:314-315), under both cfgs
        stmt = next(l for l in new_ntt.split(cfg)[1:] if re.match(r"\s*pub use [^;]*as cuda_available", l))
        for alias in ("as cuda_available", "as intt_cuda", "as ntt_cuda", "as CudaBuffer"):
            assert alias in stmt.split(";")[0], (cfg, alias)
    # every crate::ntt:: name the callers use (src/math/domain.rs:90-97,113-119) is still exported
    dom = open(dst / "src" / "math" / "domain.rs").read()
    assert dom == open(os.path.join(REFERENCE, "src", "math", "domain.rs")).read(), "the call sites are not edited"
    used = set(re.findall(r"crate::ntt::(\w+)", dom))
    assert used >= {"cuda_available", "ntt_cuda", "intt_cuda"}
    for name in used:
        assert re.search(rf"pub use [^;]*\b(?:as )?{name}\b", new_ntt), f"crate::ntt::{name} no longer resolves"
    # the reference's three GPU tests (nested in the old block, src/ntt.rs:253-311) were carried over from the checkout into `mod gpu`
    assert report["carried_test_lines"] >= 40
    gpu_mod = new_ntt[new_ntt.index("mod gpu {"):new_ntt.index("mod gpu_absent {")]
    for t in ("fn test_cuda_available", "fn test_cuda_ntt_vs_cpu", "fn test_cuda_intt_roundtrip"):
        assert t in gpu_mod and t in ref_ntt, t
    assert "use self::{gpu_available as cuda_available" in gpu_mod
    # nothing that depends on a recent prelude -- the layout guarantee is spelled as the reference spells it
    # (src/ntt.rs:115-116): `size_of` / `align_of` are prelude items only since Rust 1.80
    assert "std::mem::size_of::<BabyBear>() == std::mem::size_of::<u64>()" in gpu_mod
    assert "std::mem::align_of::<BabyBear>() == std::mem::align_of::<u64>()" in gpu_mod
    assert not re.search(r"(?<![:\w])(?:size_of|align_of)::<", new_ntt), "unqualified size_of / align_of"
    assert new_ntt.count("{") == new_ntt.count("}")
    # Cargo.toml: one [features] table, the new one; everything else as before
    cargo, ref_cargo = open(dst / "Cargo.toml").read(), open(os.path.join(REFERENCE, "Cargo.toml")).read()
    assert cargo.count("[features]") == 1 and 'hip = []' in cargo and 'cuda = ["hip"]' in cargo and "cuda = []" not in cargo
    strip = lambda t: re.sub(r"^\[features\]\n(?:(?!\[)[^\n]*\n)*", "", t, flags=re.M).split()
    assert strip(cargo) == strip(ref_cargo)
    assert open(dst / "build.rs").read() == open(os.path.join(ROOT, "rust", "build.rs")).read()
    assert not (dst / "cuda").exists() and sorted(os.listdir(dst / "hip")) == sorted(os.path.basename(f) for f in _integration_file_list())
    # applying twice is refused, not silently doubled
    with pytest.raises(SystemExit):
        _load_apply().apply(str(dst))


def test_following_integration_md_produces_a_linkable_archive(tmp_path):
    import __graft_entry__ as entry
    if os.path.isdir(REFERENCE):
        # build container: the crate IS a patched copy of the reference (rust/apply.py), so build.rs's commands run in the real layout
        crate, _ = _patched_checkout(tmp_path)
    else:
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
                          "-Wl,-Bstatic", "-ltoyni_hip", "-Wl,-Bdynamic", "-lamdhip64", "-lstdc++",
                          # not emitted by build.rs on purpose: rustc adds the platform libraries of std (-lm -lpthread -ldl ...) to every
                          # link line itself; a bare gcc link has to name them
                          "-lm", "-lpthread"],
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
