"""Host-side mirror of the reference's GPU wrapper `src/ntt.rs::cuda` (src/ntt.rs:85-315).

Same names, argument meaning and error behaviour:
  cuda_available()/gpu_available()    src/ntt.rs:144-150
  ntt_cuda(values)/ntt_gpu(values)    src/ntt.rs:224-236   in place, canonical root, no omega argument
  intt_cuda(values)/intt_gpu(values)  src/ntt.rs:239-251
  CudaBuffer/GpuBuffer                src/ntt.rs:153-215   sizes in u64 ELEMENTS
  get_or_create_ctx(n)                src/ntt.rs:128-141   process-lifetime per-n cache, never freed
`values` is a C-contiguous numpy uint64 array: the same memory layout as `&mut [BabyBear]`
(src/ntt.rs:112-116).  Errors that the reference returns as Err(String) raise ToyniError; its
assert!s raise AssertionError with the same message.
"""
import ctypes
import threading

import numpy as np

from ._lib import ToyniError, c_int, c_void_p, check, lib

_ctx_cache = {}
_ctx_lock = threading.Lock()


def gpu_available() -> bool:
    """src/ntt.rs:144-150: device-count call succeeded and count > 0."""
    count = c_int(0)
    err = lib.toyni_device_count(ctypes.byref(count))
    return err == 0 and count.value > 0


def get_or_create_ctx(n: int):
    """src/ntt.rs:128-141."""
    with _ctx_lock:
        ctx = _ctx_cache.get(n)
        if ctx is None:
            ctx = NttContext(n)
            _ctx_cache[n] = ctx
        return ctx


def _as_u64(values) -> np.ndarray:
    if not (isinstance(values, np.ndarray) and values.dtype == np.uint64 and values.flags["C_CONTIGUOUS"] and values.flags["WRITEABLE"]):
        raise TypeError("values must be a writable C-contiguous numpy uint64 array (the layout of &mut [BabyBear])")
    return values


def _checked_len(values) -> int:
    n = values.size
    assert n > 0 and n & (n - 1) == 0, "NTT size must be power of 2"  # src/ntt.rs:229
    assert n.bit_length() - 1 <= 27, "BabyBear only supports NTT up to 2^27"  # src/ntt.rs:230
    return n


def ntt_gpu(values) -> None:
    """src/ntt.rs:224-236."""
    if not gpu_available():
        raise ToyniError("GPU not available", 10004)
    v = _as_u64(values)
    n = _checked_len(v)
    get_or_create_ctx(n).run_host(v, inverse=False)


def intt_gpu(values) -> None:
    """src/ntt.rs:239-251."""
    if not gpu_available():
        raise ToyniError("GPU not available", 10004)
    v = _as_u64(values)
    n = _checked_len(v)
    get_or_create_ctx(n).run_host(v, inverse=True)


def ntt_host_multi_gpu(values: np.ndarray, n: int, devices, inverse: bool = False) -> None:
    """values: batch * n u64 elements in place; the batch is sharded contiguously over `devices` (ordinals; listing a
    device twice gives two lanes on it, whose H2D / D2H copies overlap on the full-duplex PCIe link).  One host thread
    and one context per entry, no collective (toyni_ntt_host_multi_gpu)."""
    v = _as_u64(values)
    assert v.size % n == 0
    devs = (ctypes.c_int * len(devices))(*devices)
    check(lib.toyni_ntt_host_multi_gpu(devs, len(devices), n, v.ctypes.data, v.size // n, int(inverse)), "multi-GPU host NTT failed")


EXCHANGE_PEER_COPY, EXCHANGE_RCCL = 0, 1


def ntt_slab_multi_gpu_host(values: np.ndarray, devices, inverse: bool = False, exchange: int = EXCHANGE_PEER_COPY) -> None:
    """ONE transform of values.size u64 elements (natural order, in place) over len(devices) GPUs from this process: slab pass ->
    one exchange (peer copies over xGMI, or RCCL grouped send/recv) -> relayout -> row transforms (toyni_ntt_slab_multi_gpu_host).
    A device may be listed more than once with the peer-copy exchange (lanes on one device)."""
    v = _as_u64(values)
    n = _checked_len(v)
    devs = (ctypes.c_int * len(devices))(*devices)
    check(lib.toyni_ntt_slab_multi_gpu_host(devs, len(devices), n, v.ctypes.data, int(inverse), exchange), "multi-GPU slab NTT failed")


def ntt_slab_multi_gpu_device(n: int, devices, d_slabs, d_rows, inverse: bool = False, exchange: int = EXCHANGE_PEER_COPY) -> None:
    """Device-resident form: d_slabs[g] / d_rows[g] are packed-u32 device pointers (ints) on devices[g] in the layouts of
    include/toyni_hip.h 2b; forward overwrites the slabs and fills the rows, inverse the other way round.  Blocking."""
    g = len(devices)
    assert len(d_slabs) == g and len(d_rows) == g
    devs = (ctypes.c_int * g)(*devices)
    slabs = (c_void_p * g)(*d_slabs)
    rows = (c_void_p * g)(*d_rows)
    check(lib.toyni_ntt_slab_multi_gpu_device(devs, g, n, slabs, rows, int(inverse), exchange), "multi-GPU slab NTT failed")


class NttContext:
    """Persistent per-n context: twiddles + reusable device buffers (NttCtx, cuda/ntt_kernel.cu:202-209)."""

    def __init__(self, n: int, device: int = -1):
        assert n > 0 and n & (n - 1) == 0, "NTT size must be power of 2"
        assert n.bit_length() - 1 <= 27, "BabyBear only supports NTT up to 2^27"
        h = c_void_p()
        check(lib.toyni_ntt_ctx_create(n, device, ctypes.byref(h)), "NTT context creation failed")
        self.handle = h
        self.n = n

    @property
    def passes(self) -> int:
        return lib.toyni_ntt_ctx_passes(self.handle)

    def passes_for(self, batch: int) -> int:
        """HBM sweeps one run_device call on `batch` transforms makes (the two-pass plan of 2^21, the single-sweep kernel of 2^11..2^13)."""
        return lib.toyni_ntt_ctx_passes_for(self.handle, batch)

    def set_chunk(self, chunk_elems: int) -> None:
        check(lib.toyni_ntt_ctx_set_chunk(self.handle, chunk_elems), "set_chunk failed")

    def run_host(self, values: np.ndarray, inverse: bool, batch: int = 1, shift: int = 1) -> None:
        """Host slice in place (batch * n u64 elements)."""
        v = _as_u64(values)
        assert v.size == batch * self.n, "Size mismatch"
        if shift == 1:
            check(lib.toyni_ntt_host(self.handle, v.ctypes.data, batch, int(inverse)), "GPU NTT failed")
        else:
            check(lib.toyni_coset_ntt_host(self.handle, v.ctypes.data, batch, shift, int(inverse)), "GPU coset NTT failed")

    def run_device(self, d_in: int, d_out: int, batch: int, inverse: bool, stream: int = 0, shift: int = 1) -> None:
        """Packed-u32 device pointers (ints), asynchronous on `stream` (a hipStream_t handle; 0 = HIP default stream)."""
        if shift == 1:
            check(lib.toyni_ntt_device(self.handle, d_in, d_out, batch, int(inverse), stream or None), "GPU NTT failed")
        else:
            check(lib.toyni_coset_ntt_device(self.handle, d_in, d_out, batch, shift, int(inverse), stream or None), "GPU coset NTT failed")

    def domain_elements_device(self, d_out: int, m: int, shift: int = 1, stream: int = 0) -> None:
        """d_out[i] = shift * w_m^i (roots_of_unity_domain / BabyBearDomain::elements), packed u32, m <= n."""
        check(lib.toyni_domain_elements_device(self.handle, d_out, m, shift, stream or None), "GPU domain elements failed")

    def lde_device(self, d_coeffs: int, d_out: int, batch: int, log_blowup: int, shift: int = 1, stream: int = 0) -> None:
        """Low-degree extension: forward coset transform of batch vectors of n >> log_blowup coefficients, zero padding implied."""
        check(lib.toyni_lde_device(self.handle, d_coeffs, d_out, batch, log_blowup, shift, stream or None), "GPU LDE failed")

    def lde_host(self, coeffs: np.ndarray, shift: int = 1, out: np.ndarray = None) -> np.ndarray:
        """BabyBearDomain::fft(coeffs) in one call: len(coeffs) <= n coefficients up, n evaluations on shift * <w_n> back.
        `out` (optional, n contiguous u64) is reused instead of a fresh array: from 32 MiB up the allocator hands out fresh
        mmap'ed pages per call and their first-touch faults during the download cost more than the call itself."""
        c = np.ascontiguousarray(coeffs, dtype=np.uint64)
        if out is None:
            out = np.empty(self.n, dtype=np.uint64)
        assert out.dtype == np.uint64 and out.size == self.n and out.flags.c_contiguous, "out: n contiguous u64"
        check(lib.toyni_lde_host(self.handle, c.ctypes.data if c.size else None, c.size, out.ctypes.data, shift), "GPU LDE failed")
        return out

    def lde_ext_host(self, coeffs4: np.ndarray, shift: int = 1) -> np.ndarray:
        """fft_ext of ncoeffs <= n Ext coefficients ([ncoeffs, 4] u64): [n, 4] evaluations on shift * <w_n>, padding implied."""
        c = np.ascontiguousarray(coeffs4, dtype=np.uint64).reshape(-1, 4)
        out = np.empty((self.n, 4), dtype=np.uint64)
        check(lib.toyni_lde_ext_host(self.handle, c.ctypes.data if c.size else None, c.shape[0], out.ctypes.data, shift), "GPU Ext LDE failed")
        return out

    def run_host_ext(self, values4: np.ndarray, inverse: bool, shift: int = 1) -> None:
        """n Ext elements ([n, 4] u64, AoS) in place: the four coordinate transforms as one batch, one PCIe round trip."""
        v = _as_u64(values4)
        assert v.size == 4 * self.n, "Size mismatch"
        check(lib.toyni_ntt_ext_host(self.handle, v.ctypes.data, shift, int(inverse)), "GPU Ext NTT failed")

    def run_device_ext(self, d_data: int, inverse: bool, shift: int = 1, stream: int = 0) -> None:
        check(lib.toyni_ntt_ext_device(self.handle, d_data, shift, int(inverse), stream or None), "GPU Ext NTT failed")

    def run_device_ext_batch(self, d_in: int, d_out: int, batch: int, inverse: bool, shift: int = 1, stream: int = 0) -> None:
        """fft_ext / ifft_ext of `batch` Ext vectors (packed u32 AoS, [batch][n][4]) through the interleaved passes; d_in == d_out allowed."""
        check(lib.toyni_ntt_ext_batch_device(self.handle, d_in, d_out, batch, shift, int(inverse), stream or None), "GPU Ext NTT failed")

    def lde_ext_device(self, d_coeffs: int, d_out: int, batch: int, log_blowup: int, shift: int = 1, stream: int = 0) -> None:
        """fft_ext of `batch` vectors of n >> log_blowup Ext coefficients (AoS) on the coset shift * <w_n>, zero padding implied."""
        check(lib.toyni_lde_ext_batch_device(self.handle, d_coeffs, d_out, batch, log_blowup, shift, stream or None), "GPU Ext LDE failed")

    def run_device_u64(self, d_data: int, batch: int, inverse: bool, stream: int = 0) -> None:
        check(lib.toyni_ntt_device_u64(self.handle, d_data, batch, int(inverse), stream or None), "GPU NTT failed")

    # ---- measurement build only (libtoyni_hip_tools.so via TOYNI_LIB_OVERRIDE; include/toyni_hip_tools.h) ----
    @staticmethod
    def _need_tools():
        from . import _lib
        if not _lib.HAS_TOOLS:
            raise RuntimeError("launch timing hooks exist only in the measurement build: build libtoyni_hip_tools.so "
                               "(__graft_entry__.build_tools()) and point TOYNI_LIB_OVERRIDE at it")

    def profile_passes(self, d_data: int, batch: int, inverse: bool, reps: int = 20, stream: int = 0):
        """Average launch duration (ms) of each pass kernel, HIP events on `stream`."""
        self._need_tools()
        out = (ctypes.c_float * 3)()
        check(lib.toyni_ntt_profile_passes(self.handle, d_data, batch, int(inverse), reps, out, stream or None), "profile failed")
        return [out[i] for i in range(self.passes_for(batch) if self.passes == 3 else self.passes)]   # (the pass kernels, not the single-sweep one)

    def timing(self, enable: bool) -> None:
        """Bracket every pass launch of this context with HIP events on its launch stream (see read_timing)."""
        self._need_tools()
        check(lib.toyni_ntt_ctx_timing(self.handle, int(enable)), "timing switch failed")

    def read_timing(self):
        """{'forward': [avg ms per launch of pass 0, ...], 'inverse': [...], 'launches': {...}} for the launches since timing(True)."""
        self._need_tools()
        ms = (ctypes.c_float * 6)()
        cnt = (ctypes.c_uint32 * 6)()
        check(lib.toyni_ntt_ctx_timing_read(self.handle, ms, cnt), "timing read failed")
        out = {"launches": {}}
        for d, name in enumerate(("forward", "inverse")):
            out[name] = [ms[3 * d + p] / cnt[3 * d + p] if cnt[3 * d + p] else None for p in range(self.passes)]
            out["launches"][name] = [int(cnt[3 * d + p]) for p in range(self.passes)]
        return out

    def trim(self) -> None:
        """Device-wide synchronisation, then every intermediate buffer of the context is freed."""
        check(lib.toyni_ntt_ctx_trim(self.handle), "trim failed")

    def synchronize(self, stream: int = 0) -> None:
        check(lib.toyni_stream_synchronize(self.handle, stream or None), "stream synchronize failed")

    def destroy(self) -> None:
        if self.handle:
            lib.toyni_ntt_ctx_destroy(self.handle)
            self.handle = None


class PinnedArray:
    """count u64 elements of pinned (page-locked) host memory as a numpy array (`.array`), from toyni_host_alloc.  Host-slice
    calls on it (run_host, ntt_host_multi_gpu) take the pipelined path: upload, kernels and download overlap."""

    def __init__(self, count: int):
        p = c_void_p()
        check(lib.toyni_host_alloc(ctypes.byref(p), count * 8), "pinned host allocation failed")
        self.ptr = p
        self.array = np.ctypeslib.as_array((ctypes.c_uint64 * count).from_address(p.value))

    def free(self) -> None:
        if self.ptr:
            self.array = None
            lib.toyni_host_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class GpuBuffer:
    """RAII device buffer of `size` u64 elements (CudaBuffer, src/ntt.rs:153-215)."""

    def __init__(self, size: int):
        p = c_void_p()
        err = lib.cuda_malloc(ctypes.byref(p), size)
        if err != 0:
            raise ToyniError("GPU malloc failed", err)  # src/ntt.rs:163-166
        self.ptr = p
        self.size = size

    def copy_from_host(self, data: np.ndarray) -> None:
        assert data.size == self.size, "Size mismatch"  # src/ntt.rs:172
        d = np.ascontiguousarray(data, dtype=np.uint64)
        err = lib.cuda_copy_to_device(self.ptr, d.ctypes.data, self.size)
        if err != 0:
            raise ToyniError("GPU copy to device failed", err)

    def copy_to_host(self, data: np.ndarray) -> None:
        assert data.size == self.size, "Size mismatch"  # src/ntt.rs:187
        v = _as_u64(data)
        err = lib.cuda_copy_from_device(v.ctypes.data, self.ptr, self.size)
        if err != 0:
            raise ToyniError("GPU copy from device failed", err)

    def as_ptr(self) -> int:
        return self.ptr.value

    def free(self) -> None:
        if self.ptr:
            lib.cuda_free(self.ptr)
            self.ptr = None

    def __del__(self):  # Drop, src/ntt.rs:206-212
        try:
            self.free()
        except Exception:
            pass


# the reference's names (Cargo feature `cuda`; src/ntt.rs:314-315) resolve to the same objects
cuda_available = gpu_available
ntt_cuda = ntt_gpu
intt_cuda = intt_gpu
CudaBuffer = GpuBuffer
