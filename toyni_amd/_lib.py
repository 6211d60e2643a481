"""ctypes binding of libtoyni_hip.so (include/toyni_hip.h).  No fallback: a missing library is an error."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TOYNI_LIB_OVERRIDE") or os.path.join(_HERE, "lib", "libtoyni_hip.so")  # override: diagnostic builds

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} not found: the HIP backend is not built. Run `python -c 'import __graft_entry__ as g; g.build()'` "
        "(hipcc --offload-arch=gfx950). toyni_amd has no CPU path."
    )



def _one_hip_runtime_per_process():
    """libtoyni_hip.so needs libamdhip64.so.7.  A PyTorch-ROCm wheel bundles its OWN copy of that runtime (same
    SONAME, different file); if ours binds to /opt/rocm's copy first and torch is imported later, the process ends up
    with two HIP/HSA runtimes and the second one finds no GPU.  So when torch is installed, map its runtime first:
    the dynamic loader then resolves our DT_NEEDED to it by SONAME, and a later `import torch` finds its own file
    already mapped.  TOYNI_HIP_RUNTIME=system skips this (e.g. torch-free profiling drivers)."""
    import importlib.util
    import sys
    if os.environ.get("TOYNI_HIP_RUNTIME", "auto") == "system" or "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")  # locates the package, does not import it
    except (ImportError, ValueError):
        spec = None
    if spec is not None and spec.origin:
        cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
        if os.path.exists(cand):
            ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)


_one_hip_runtime_per_process()
lib = ctypes.CDLL(LIB_PATH)

c_int = ctypes.c_int
c_u32 = ctypes.c_uint32
c_u64 = ctypes.c_uint64
c_size = ctypes.c_size_t
c_void_p = ctypes.c_void_p
p_u64 = ctypes.POINTER(ctypes.c_uint64)
p_u32 = ctypes.POINTER(ctypes.c_uint32)

# toyni_fri_challenge_fn: int (*)(void* user, unsigned round, const uint8_t* prev_root32, uint32_t* beta_out)
FRI_CHALLENGE_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_uint, ctypes.POINTER(ctypes.c_uint8), ctypes.POINTER(ctypes.c_uint32))

# name -> (restype, argtypes); mirrors include/toyni_hip.h line by line
SIGNATURES = {
    # section 1: the reference's ABI
    "ntt_ctx_create": (c_void_p, [c_u32]),
    "ntt_ctx_destroy": (None, [c_void_p]),
    "ntt_run_inplace": (None, [c_void_p, c_void_p]),
    "intt_run_inplace": (None, [c_void_p, c_void_p]),
    "cuda_malloc": (c_int, [ctypes.POINTER(c_void_p), c_size]),
    "cuda_free": (c_int, [c_void_p]),
    "cuda_copy_to_device": (c_int, [c_void_p, c_void_p, c_size]),
    "cuda_copy_from_device": (c_int, [c_void_p, c_void_p, c_size]),
    "cuda_get_error_string": (ctypes.c_char_p, [c_int]),
    "cudaGetDeviceCount": (c_int, [ctypes.POINTER(c_int)]),
    "toyni_device_count": (c_int, [ctypes.POINTER(c_int)]),
    "toyni_error_string": (ctypes.c_char_p, [c_int]),
    # section 2
    "toyni_ntt_ctx_create": (c_int, [c_u32, c_int, ctypes.POINTER(c_void_p)]),
    "toyni_ntt_ctx_destroy": (c_int, [c_void_p]),
    "toyni_ntt_ctx_n": (c_u32, [c_void_p]),
    "toyni_ntt_ctx_device": (c_int, [c_void_p]),
    "toyni_ntt_ctx_passes": (c_int, [c_void_p]),
    "toyni_ntt_ctx_passes_for": (c_int, [c_void_p, c_size]),
    "toyni_ntt_ctx_set_chunk": (c_int, [c_void_p, c_size]),
    "toyni_ntt_host": (c_int, [c_void_p, c_void_p, c_size, c_int]),
    "toyni_ntt_host_multi_gpu": (c_int, [c_void_p, c_int, c_u32, c_void_p, c_size, c_int]),
    "toyni_ntt_device": (c_int, [c_void_p, c_void_p, c_void_p, c_size, c_int, c_void_p]),
    "toyni_ntt_device_u64": (c_int, [c_void_p, c_void_p, c_size, c_int, c_void_p]),
    "toyni_coset_ntt_device": (c_int, [c_void_p, c_void_p, c_void_p, c_size, c_u32, c_int, c_void_p]),
    "toyni_coset_ntt_host": (c_int, [c_void_p, c_void_p, c_size, c_u64, c_int]),
    "toyni_lde_device": (c_int, [c_void_p, c_void_p, c_void_p, c_size, ctypes.c_uint, c_u32, c_void_p]),
    "toyni_domain_elements_device": (c_int, [c_void_p, c_void_p, c_size, c_u32, c_void_p]),
    "toyni_lde_host": (c_int, [c_void_p, c_void_p, c_size, c_void_p, c_u64]),
    "toyni_lde_ext_host": (c_int, [c_void_p, c_void_p, c_size, c_void_p, c_u64]),
    "toyni_lde_ext_device": (c_int, [c_void_p, c_void_p, c_void_p, ctypes.c_uint, c_u32, c_void_p]),
    "toyni_ntt_ext_host": (c_int, [c_void_p, c_void_p, c_u64, c_int]),
    "toyni_ntt_ext_device": (c_int, [c_void_p, c_void_p, c_u32, c_int, c_void_p]),
    "toyni_ntt_ext_batch_device": (c_int, [c_void_p, c_void_p, c_void_p, c_size, c_u32, c_int, c_void_p]),
    "toyni_lde_ext_batch_device": (c_int, [c_void_p, c_void_p, c_void_p, c_size, ctypes.c_uint, c_u32, c_void_p]),
    "toyni_fourstep_twiddle_device": (c_int, [c_void_p, c_void_p, c_size, c_size, c_size, c_int, c_void_p]),
    "toyni_ntt_ctx_first_pass_points": (c_size, [c_void_p]),
    "toyni_first_pass_points": (c_size, [c_u32]),
    "toyni_ntt_slab_pass_device": (c_int, [c_void_p, c_void_p, c_size, c_size, c_int, c_void_p]),
    "toyni_ntt_slab_relayout_device": (c_int, [c_void_p, c_void_p, c_void_p, c_size, c_size, c_size, c_int, c_void_p]),
    "toyni_ntt_slab_rows_device": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_size, c_size, c_size, c_int, ctypes.POINTER(c_int), c_void_p]),
    "toyni_ntt_slab_multi_gpu_device": (c_int, [c_void_p, c_int, c_u32, c_void_p, c_void_p, c_int, c_int]),
    "toyni_ntt_slab_multi_gpu_host": (c_int, [c_void_p, c_int, c_u32, c_void_p, c_int, c_int]),
    # section 3
    "toyni_fri_fold_device": (c_int, [c_void_p, c_void_p, c_void_p, c_size, c_u32, c_u32, c_void_p]),
    "toyni_fri_fold_layers_device": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, ctypes.c_uint, c_u32, c_void_p]),
    "toyni_fri_fold_xs_device": (c_int, [c_void_p, c_void_p, c_void_p, c_size, c_u32, c_void_p]),
    "toyni_fri_fold_host": (c_int, [c_void_p, c_void_p, c_size, c_void_p, c_u64]),
    "toyni_fri_fold_ext_device": (c_int, [c_void_p, c_void_p, c_void_p, c_size, c_void_p, c_u32, c_void_p]),
    "toyni_fri_fold_ext_xs_device": (c_int, [c_void_p, c_void_p, c_void_p, c_size, c_void_p, c_void_p]),
    "toyni_fri_fold_ext_host": (c_int, [c_void_p, c_void_p, c_size, c_void_p, c_void_p]),
    "toyni_merkle_total_digests": (c_size, [c_size]),
    "toyni_merkle_commit_device": (c_int, [c_void_p, c_void_p, c_size, c_void_p, c_void_p]),
    "toyni_merkle_commit_host": (c_int, [c_void_p, c_void_p, c_size, c_void_p]),
    # section 3c
    "toyni_fri_fold_commit_device": (c_int, [c_void_p, c_void_p, c_void_p, c_size, c_u32, c_u32, c_void_p, c_void_p, c_void_p]),
    "toyni_fri_commit_phase_device": (c_int, [c_void_p, c_void_p, c_size, c_u32, c_size, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                              c_void_p, ctypes.POINTER(ctypes.c_uint), c_void_p]),
    "toyni_fib_quotient_device": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, ctypes.c_uint, c_u32, c_void_p]),
    "toyni_fib_deep_device": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, ctypes.c_uint, c_u32, c_u32, c_void_p, c_void_p]),
    "toyni_poly_eval_device": (c_int, [c_void_p, c_void_p, c_size, c_void_p, ctypes.c_uint, c_void_p, c_void_p]),
    "toyni_merkle_open_record_bytes": (c_size, [c_size]),
    "toyni_merkle_open_device": (c_int, [c_void_p, c_size, c_void_p, c_void_p, c_void_p, c_size, c_void_p, c_void_p]),
    "toyni_merkle_open_groups_device": (c_int, [c_void_p, c_size, c_void_p]),
    # section 4
    "toyni_malloc": (c_int, [ctypes.POINTER(c_void_p), c_size]),
    "toyni_free": (c_int, [c_void_p]),
    "toyni_host_alloc": (c_int, [ctypes.POINTER(c_void_p), c_size]),
    "toyni_host_free": (c_int, [c_void_p]),
    "toyni_memcpy_h2d": (c_int, [c_void_p, c_void_p, c_size]),
    "toyni_memcpy_d2h": (c_int, [c_void_p, c_void_p, c_size]),
    "toyni_memcpy_h2d_async": (c_int, [c_void_p, c_void_p, c_size, c_void_p]),
    "toyni_memcpy_d2h_async": (c_int, [c_void_p, c_void_p, c_size, c_void_p]),
    "toyni_memcpy_d2d_async": (c_int, [c_void_p, c_void_p, c_size, c_void_p]),
    "toyni_memset_async": (c_int, [c_void_p, c_int, c_size, c_void_p]),
    "toyni_chacha20_fill_device": (c_int, [c_void_p, c_size, c_void_p, c_u64, c_void_p]),
    "toyni_narrow_u64_to_u32": (c_int, [c_void_p, c_void_p, c_size, c_void_p]),
    "toyni_widen_u32_to_u64": (c_int, [c_void_p, c_void_p, c_size, c_void_p]),
    "toyni_stream_create": (c_int, [ctypes.POINTER(c_void_p), c_int]),
    "toyni_stream_destroy": (c_int, [c_void_p]),
    "toyni_stream_wait": (c_int, [c_void_p, c_void_p]),
    "toyni_event_create": (c_int, [ctypes.POINTER(c_void_p)]),
    "toyni_event_destroy": (c_int, [c_void_p]),
    "toyni_event_record": (c_int, [c_void_p, c_void_p]),
    "toyni_stream_wait_event": (c_int, [c_void_p, c_void_p]),
    "toyni_stream_synchronize": (c_int, [c_void_p, c_void_p]),
    "toyni_ntt_ctx_trim": (c_int, [c_void_p]),
    "toyni_set_device": (c_int, [c_int]),
    "toyni_launched_kernels": (c_size, [ctypes.c_char_p, c_size]),
}

# include/toyni_hip_tools.h: present only in the measurement build libtoyni_hip_tools.so (TOYNI_LIB_OVERRIDE points at it)
TOOLS_SIGNATURES = {
    "toyni_ntt_ctx_timing": (c_int, [c_void_p, c_int]),
    "toyni_ntt_ctx_timing_read": (c_int, [c_void_p, c_void_p, c_void_p]),
    "toyni_ntt_profile_passes": (c_int, [c_void_p, c_void_p, c_size, c_int, c_int, ctypes.POINTER(ctypes.c_float), c_void_p]),
    "toyni_tools_inject": (c_int, [ctypes.c_uint]),
    "toyni_tools_rccl_version": (c_int, []),
    "toyni_tools_slab_phases": (c_int, [c_int]),
    "toyni_tools_slab_phases_read": (c_int, [ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_uint)]),
}

for _name, (_res, _args) in SIGNATURES.items():
    _f = getattr(lib, _name)  # AttributeError here = the library does not export what the header declares
    _f.restype = _res
    _f.argtypes = _args
HAS_TOOLS = hasattr(lib, "toyni_ntt_ctx_timing")
if HAS_TOOLS:
    for _name, (_res, _args) in TOOLS_SIGNATURES.items():
        _f = getattr(lib, _name)
        _f.restype = _res
        _f.argtypes = _args


def error_string(status: int) -> str:
    return lib.toyni_error_string(status).decode()


class ToyniError(RuntimeError):
    """Raised where the reference returns Err(String) (src/ntt.rs:163-166 pattern)."""

    def __init__(self, what: str, status: int):
        super().__init__(f"{what}: {error_string(status)}")
        self.status = status


def check(status: int, what: str) -> None:
    if status != 0:
        raise ToyniError(what, status)


def launched_kernels():
    """Symbols of the kernels this process has launched through the library so far (diagnostics; toyni_launched_kernels)."""
    need = lib.toyni_launched_kernels(None, 0)
    buf = ctypes.create_string_buffer(need + 4096)
    lib.toyni_launched_kernels(buf, len(buf))
    return [s for s in buf.value.decode().split("\n") if s]
