"""The C-ABI library loads and exports every symbol include/toyni_hip.h declares; host-side argument
checks behave like the reference's asserts.  No compute (no GPU here)."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as entry
    entry.build_hip()
    from toyni_amd import _lib
    return _lib


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "toyni_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\(", src)
    return sorted({n for n in names if n.startswith(("toyni_", "ntt_", "intt_", "cuda_", "cudaGet"))})


def test_header_symbols_all_exported(lib):
    declared = _declared_symbols()
    assert len(declared) >= 35
    for name in declared:
        assert hasattr(lib.lib, name), f"{name} declared in include/toyni_hip.h but not exported"
    # and the binding table covers the header exactly
    assert sorted(lib.SIGNATURES) == declared


def test_measurement_hooks_live_in_the_tools_build_only(lib):
    """include/toyni_hip_tools.h is exported by libtoyni_hip_tools.so (the same source with -DTOYNI_TOOLS) and by nothing shipped."""
    import ctypes
    import __graft_entry__ as entry
    src = open(os.path.join(ROOT, "include", "toyni_hip_tools.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    declared = sorted(set(re.findall(r"\b(toyni_[a-z0-9_]*)\s*\(", src)))
    assert declared == sorted(lib.TOOLS_SIGNATURES)
    tools = ctypes.CDLL(entry.build_tools())
    for name in declared:
        assert hasattr(tools, name) and not hasattr(lib.lib, name), name
    for name in lib.SIGNATURES:                      # the measurement build is a superset of the shipped ABI
        assert hasattr(tools, name), name
    assert not hasattr(lib.lib, "toyni_microbench") and not hasattr(tools, "toyni_microbench")   # its own program: tools/microbench.hip


def test_reference_abi_names_present(lib):
    # all ten symbols src/ntt.rs:95-110 binds, cudaGetDeviceCount (libcudart's in the reference) included: the reference's extern
    # block links with the link name as its only edit; toyni_device_count is the same call under a neutral name
    for name in ["ntt_ctx_create", "ntt_ctx_destroy", "ntt_run_inplace", "intt_run_inplace", "cuda_malloc", "cuda_free",
                 "cuda_copy_to_device", "cuda_copy_from_device", "cuda_get_error_string", "cudaGetDeviceCount", "toyni_device_count"]:
        assert hasattr(lib.lib, name)
    import ctypes
    a, b = ctypes.c_int(-1), ctypes.c_int(-2)
    ra, rb = lib.lib.cudaGetDeviceCount(ctypes.byref(a)), lib.lib.toyni_device_count(ctypes.byref(b))
    assert (ra, a.value) == (rb, b.value)


def test_error_strings(lib):
    assert lib.error_string(0) == "success"
    assert "power of two" in lib.error_string(10001)
    assert lib.error_string(10003) == "Evaluations length must be even"  # src/math/fri.rs:28
    assert lib.error_string(10005) == "Cannot invert zero"               # src/babybear.rs:112


def test_size_validation_without_gpu(lib):
    import ctypes
    h = ctypes.c_void_p()
    assert lib.lib.toyni_ntt_ctx_create(3, -1, ctypes.byref(h)) == 10001        # not a power of two
    assert lib.lib.toyni_ntt_ctx_create(0, -1, ctypes.byref(h)) == 10001
    assert lib.lib.toyni_ntt_ctx_create(1 << 28, -1, ctypes.byref(h)) == 10001  # > 2^27 (src/ntt.rs:230)
    assert lib.lib.ntt_ctx_create(1 << 28) is None                               # cuda/ntt_kernel.cu:217-220
    assert lib.lib.toyni_ntt_ctx_destroy(None) == 0                              # null-safe like :237
    assert lib.lib.toyni_fri_fold_host(None, None, 4, None, 1) == 10002


def test_host_mirror_asserts_like_reference():
    import toyni_amd
    if toyni_amd.gpu_available():
        pytest.skip("CPU-side behaviour test")
    v = np.zeros(8, dtype=np.uint64)
    with pytest.raises(toyni_amd._lib.ToyniError):   # src/ntt.rs:225-227 Err("CUDA not available")
        toyni_amd.ntt_cuda(v)
    with pytest.raises(AssertionError, match="even"):
        toyni_amd.fri_fold(np.zeros(3, dtype=np.uint64), np.ones(3, dtype=np.uint64), 1)
    with pytest.raises(NotImplementedError):
        toyni_amd.BabyBearDomain(8).fft([1, 2, 3])   # no CPU path in this package


def _c_prototypes():
    """name -> number of parameters, from include/toyni_hip.h"""
    src = open(os.path.join(ROOT, "include", "toyni_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    protos = {}
    for m in re.finditer(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\(([^;{}]*?)\)\s*;", src, flags=re.S):
        name, args = m.group(1), m.group(2).strip()
        if name.startswith(("toyni_", "ntt_", "intt_", "cuda_", "cudaGet")):
            protos[name] = 0 if args in ("", "void") else args.count(",") + 1
    return protos


def test_rust_and_integration_bindings_match_the_header():
    # No Rust toolchain exists here, so at least keep the shipped extern blocks honest: every function the
    # Rust binding (rust/src/ntt_gpu.rs) and INTEGRATION.md declare must exist in the header with the same arity.
    protos = _c_prototypes()
    decl = re.compile(r"\bfn\s+([A-Za-z_][A-Za-z0-9_]*)\s*\(([^)]*)\)\s*(?:->\s*[^;{]+)?;")
    checked = 0
    for path in ("rust/src/ntt_gpu.rs", "INTEGRATION.md"):
        text = open(os.path.join(ROOT, path)).read()
        for name, args in decl.findall(text):
            if not name.startswith(("toyni_", "ntt_", "intt_", "cuda_", "cudaGet")):
                continue
            assert name in protos, f"{path}: {name} is not declared in include/toyni_hip.h"
            arity = 0 if not args.strip() else args.count(",") + 1
            assert arity == protos[name], f"{path}: {name} takes {arity} arguments, the header says {protos[name]}"
            checked += 1
    assert checked >= 15


def _rust_exports(cfg_marker):
    """Names `rust/src/ntt_gpu.rs` re-exports at file scope (i.e. as `crate::ntt::<name>` once pasted into src/ntt.rs) under the
    `#[cfg(...)]` line containing `cfg_marker`, and the names the module they come from really defines as pub / pub(crate)."""
    text = open(os.path.join(ROOT, "rust", "src", "ntt_gpu.rs")).read()
    exported, defined = set(), set()
    for m in re.finditer(r"#\[cfg\(([^\]]*)\)\]\s*\n(pub(?:\(crate\))? use (\w+)::\{([^}]*)\};)", text):
        if m.group(1).replace(" ", "") != cfg_marker.replace(" ", ""):
            continue
        module = m.group(3)
        body = re.search(r"mod %s \{(.*?)\n\}\n" % module, text, flags=re.S).group(1)
        for item in m.group(4).split(","):
            parts = item.split(" as ")
            src_name, out_name = parts[0].strip(), parts[-1].strip()
            assert re.search(r"pub(?:\(crate\))? (?:fn|struct) %s\b" % src_name, body), f"{module}::{src_name} is re-exported but not pub in the module"
            exported.add(out_name)
        defined |= set(re.findall(r"pub(?:\(crate\))? (?:fn|struct) (\w+)", body))
    return exported


def test_integration_sketches_only_use_names_the_rust_module_exports():
    """INTEGRATION.md's sketches called `crate::ntt::context`, which was private.  Without rustc, check the
    next best thing: every `crate::ntt::<name>` that INTEGRATION.md uses and every public name of the reference's GPU surface
    (src/ntt.rs:314-315 + CudaBuffer) is exported by rust/src/ntt_gpu.rs both with hipcc (`has_hip`) and in the stub build."""
    used = set(re.findall(r"crate::ntt::(\w+)", open(os.path.join(ROOT, "INTEGRATION.md")).read()))
    assert {"context", "gpu_available"} <= used
    reference_surface = {"cuda_available", "ntt_cuda", "intt_cuda", "CudaBuffer"}
    neutral = {"gpu_available", "ntt_gpu", "intt_gpu", "GpuBuffer"}
    for cfg in ('all(feature = "hip", has_hip)', 'all(feature = "hip", not(has_hip))'):
        exported = _rust_exports(cfg)
        missing = (used | reference_surface | neutral) - exported
        assert not missing, f"rust/src/ntt_gpu.rs under cfg({cfg}) does not export {sorted(missing)}"
    # the stub's GpuBuffer has the real one's methods (a user of CudaBuffer must compile without hipcc)
    text = open(os.path.join(ROOT, "rust", "src", "ntt_gpu.rs")).read()
    real = re.search(r"mod gpu \{(.*?)\n\}\n", text, flags=re.S).group(1)
    stub = re.search(r"mod gpu_absent \{(.*?)\n\}\n", text, flags=re.S).group(1)
    methods = lambda body: set(re.findall(r"pub fn (\w+)\(", body[body.index("impl GpuBuffer"):]))
    assert methods(real) == methods(stub) == {"new", "copy_from_host", "copy_to_host", "as_ptr"}
    # and INTEGRATION.md says which feature the reference's own call sites need
    integ = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    assert "--features cuda" in integ and '`--features hip` alone' in integ


def test_hand_declared_rccl_abi_matches_rccl_h():
    """csrc/multi_gpu.hpp binds librccl with dlopen and declares the five entry points by hand (so that a crate's hip/ directory
    builds without RCCL's include path).  ADVICE r2: those declarations had never met the real header.  Parse rccl.h and check the
    enum value of ncclUint32, the parameter lists of the functions used, and that the header's version satisfies the minimum."""
    hdr = "/opt/rocm/include/rccl/rccl.h"
    if not os.path.exists(hdr):
        pytest.skip("no rccl.h in this image")
    text = open(hdr).read()
    src = open(os.path.join(ROOT, "toyni_amd", "csrc", "multi_gpu.hpp")).read()
    assert re.search(r"ncclUint32\s*=\s*3\b", text) and re.search(r"constexpr int RCCL_UINT32 = 3;", src)
    norm = lambda t: re.sub(r"\s+", " ", t).strip()

    def params(name):
        m = re.search(r"ncclResult_t\s+%s\s*\(([^)]*)\)\s*;" % name, text)
        assert m, name
        out = []
        for prm in m.group(1).split(","):
            prm = norm(prm)
            prm = re.sub(r"\b\w+$", "", prm).strip() if not prm.endswith("*") else prm   # drop the parameter name
            out.append(prm.replace(" *", "*").replace("* ", "*"))
        return out
    assert params("ncclCommInitAll") == ["ncclComm_t*", "int", "const int*"]
    assert params("ncclSend") == ["const void*", "size_t", "ncclDataType_t", "int", "ncclComm_t", "hipStream_t"]
    assert params("ncclRecv") == ["void*", "size_t", "ncclDataType_t", "int", "ncclComm_t", "hipStream_t"]
    assert params("ncclGetVersion") == ["int*"]
    assert re.search(r"ncclResult_t\s+ncclGroupStart\s*\(\s*(void)?\s*\)", text) and re.search(r"ncclResult_t\s+ncclGroupEnd\s*\(\s*(void)?\s*\)", text)
    assert re.search(r"typedef struct ncclComm\s*\*\s*ncclComm_t;", text)          # a pointer: `typedef void* rccl_comm_t` has its size
    # and the hand-written side says the same
    assert "int (*CommInitAll)(rccl_comm_t*, int, const int*)" in src
    assert "int (*Send)(const void*, size_t, int, int, rccl_comm_t, hipStream_t)" in src
    assert "int (*Recv)(void*, size_t, int, int, rccl_comm_t, hipStream_t)" in src
    code = int(re.search(r"#define NCCL_VERSION_CODE (\d+)", text).group(1))
    assert code >= int(re.search(r"RCCL_MIN_VERSION = (\d+);", src).group(1))
