"""GPU parity tests: the HIP path, called through the C ABI (ctypes over include/toyni_hip.h), against
the oracle and the committed golden vectors -- bit-exact (integer path, no tolerance).

The first three tests are the reference's own GPU tests (src/ntt.rs:253-311) restated; the rest widen
them: every size 2^0..2^22, the golden vectors, edge vectors, ragged batches, the device-resident /
u64 / coset entry points, the FRI fold in all its forms, and full-size (2^24, 2^27, 1024 x 2^20)
runs checked through size-independent properties."""
import ctypes

import numpy as np
import pytest

import oracle
from oracle import P

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ta():
    import __graft_entry__ as entry
    entry.build_hip()
    # torch (used by the 4-step tests for device tensors) brings its own copy of the HIP runtime: initialise it FIRST,
    # as bench.py does, so that the whole session runs on one runtime whose device state was set up once, up front
    import torch
    assert torch.cuda.is_available(), "GPU tests need a device"
    torch.cuda.init()
    import toyni_amd
    assert toyni_amd.gpu_available(), "GPU tests need a device"
    return toyni_amd


class DevBuf:
    """Raw device allocation through the ABI's plumbing calls."""

    def __init__(self, ta, nbytes):
        self.lib = ta._lib.lib
        p = ctypes.c_void_p()
        ta._lib.check(self.lib.toyni_malloc(ctypes.byref(p), max(nbytes, 4)), "malloc")
        self.ptr = p.value
        self.nbytes = nbytes

    def upload(self, arr, offset=0):
        arr = np.ascontiguousarray(arr)
        assert self.lib.toyni_memcpy_h2d(self.ptr + offset, arr.ctypes.data, arr.nbytes) == 0

    def download(self, dtype, count, offset=0):
        out = np.empty(count, dtype=dtype)
        assert self.lib.toyni_memcpy_d2h(out.ctypes.data, self.ptr + offset, out.nbytes) == 0
        return out

    def free(self):
        if self.ptr:
            self.lib.toyni_free(self.ptr)
            self.ptr = None


def dev_transform(ta, x32, n, batch, inverse, inplace=True, shift=1, chunk=None):
    ctx = ta.ntt.get_or_create_ctx(n)
    if chunk is not None:
        ctx.set_chunk(chunk)
    a = DevBuf(ta, x32.nbytes)
    b = a if inplace else DevBuf(ta, x32.nbytes)
    try:
        a.upload(x32)
        ctx.run_device(a.ptr, b.ptr, batch, inverse, shift=shift)
        ctx.synchronize()
        return b.download(np.uint32, x32.size)
    finally:
        a.free()
        if b is not a:
            b.free()
        if chunk is not None:
            ctx.set_chunk(0)


# ---------------------------------------------------------------- reference tests, src/ntt.rs:253-311
def test_cuda_available(ta):
    assert ta.cuda_available() is True


def test_cuda_ntt_vs_cpu(ta):
    n = 256
    cpu_values = oracle.pattern_7i3(n)               # (i * 7 + 3), src/ntt.rs:272
    gpu_values = cpu_values.copy()
    omega = oracle.root_of_unity(8)
    cpu_values = oracle.ntt(cpu_values, omega)       # cpu_ntt
    ta.ntt_cuda(gpu_values)
    for i, (c, g) in enumerate(zip(cpu_values, gpu_values)):
        assert c == g, f"Mismatch at index {i}: CPU={c}, GPU={g}"


def test_cuda_intt_roundtrip(ta):
    n = 256
    original = oracle.pattern_7i3(n)
    values = original.copy()
    ta.ntt_cuda(values)
    ta.intt_cuda(values)
    assert (values == original).all(), "Roundtrip failed"


# ---------------------------------------------------------------- golden vectors
def test_ntt_golden(ta, golden):
    for c in golden["ntt"]:
        v = np.array(c["input"], dtype=np.uint64)
        ta.ntt_gpu(v)
        assert v.tolist() == c["forward"], c["name"]
        v = np.array(c["input"], dtype=np.uint64)
        ta.intt_gpu(v)
        assert v.tolist() == c["inverse"], c["name"]


def test_kat_n8_polynomial_evaluation(ta):
    # src/ntt.rs:338-357
    v = np.array([1, 2, 3, 0, 0, 0, 0, 0], dtype=np.uint64)
    ta.ntt_gpu(v)
    assert v[0] == 6
    x = oracle.root_of_unity(3)
    assert v[1] == (1 + 2 * x + 3 * x * x) % P


# ---------------------------------------------------------------- every size vs the oracle
@pytest.mark.parametrize("log_n", list(range(0, 23)))
def test_host_path_all_sizes(ta, log_n):
    n = 1 << log_n
    x = oracle.splitmix(n, 0x70796E69 + (log_n << 32))
    v = x.copy()
    ta.ntt_gpu(v)
    assert (v == oracle.ntt(x)).all(), f"forward n=2^{log_n}"
    v = x.copy()
    ta.intt_gpu(v)
    assert (v == oracle.intt(x)).all(), f"inverse n=2^{log_n}"


@pytest.mark.parametrize("log_n", [20, 24])
def test_headline_sizes_pattern_and_random(ta, log_n):
    # BASELINE configs[1]: forward + inverse at n = 2^20 (and 2^24), bit-exact vs the CPU path, inputs I1 + I2
    n = 1 << log_n
    for x in (oracle.pattern_7i3(n), oracle.splitmix(n, 99)):
        want = oracle.ntt(x)
        got = dev_transform(ta, x.astype(np.uint32), n, 1, False)
        assert (got == want).all()
        back = dev_transform(ta, got, n, 1, True, inplace=False)
        assert (back == x).all()


def test_edge_vectors_n20(ta):
    n = 1 << 20
    ctx_root = oracle.root_of_unity(20)
    zeros = np.zeros(n, dtype=np.uint32)
    assert not dev_transform(ta, zeros, n, 1, False).any()
    d0 = zeros.copy(); d0[0] = 1
    assert (dev_transform(ta, d0, n, 1, False) == 1).all()                       # delta_0 -> all ones
    d1 = zeros.copy(); d1[1] = 1
    assert (dev_transform(ta, d1, n, 1, False) == oracle.roots_of_unity_domain(n)).all()  # delta_1 -> w^k
    c = np.full(n, 123456789, dtype=np.uint32)
    out = dev_transform(ta, c, n, 1, False)
    assert out[0] == (123456789 * n) % P and not out[1:].any()                   # constant -> n c at 0
    pm1 = np.full(n, P - 1, dtype=np.uint32)
    out = dev_transform(ta, pm1, n, 1, False)
    assert out[0] == ((P - 1) * n) % P and not out[1:].any()
    assert ctx_root == 195061667


# ---------------------------------------------------------------- batches, chunks, entry-point forms
# (20, 5) runs the 1024-point passes on 16-wide tiles, (20, 18) on 32-wide, (19, 3) / batch 1 on 8-wide
@pytest.mark.parametrize("log_n,batch", [(1, 1), (3, 70), (5, 257), (6, 33), (8, 19), (10, 9), (11, 5), (16, 5), (19, 3), (20, 5), (20, 18), (21, 2)])
def test_batched_device_resident(ta, log_n, batch):
    n = 1 << log_n
    x = oracle.splitmix(n * batch, 7 + log_n).reshape(batch, n)
    want = np.stack([oracle.ntt(r) for r in x])
    got = dev_transform(ta, x.astype(np.uint32).reshape(-1), n, batch, False, inplace=False).reshape(batch, n)
    assert (got == want).all()
    back = dev_transform(ta, got.reshape(-1), n, batch, True).reshape(batch, n)
    assert (back == x).all()


def test_chunked_batch_equals_unchunked(ta):
    n, batch = 1 << 12, 13
    x = oracle.splitmix(n * batch, 5).astype(np.uint32)
    a = dev_transform(ta, x, n, batch, False)
    b = dev_transform(ta, x, n, batch, False, chunk=3 * n)      # 3 transforms per chunk, ragged tail
    c = dev_transform(ta, x, n, batch, False, chunk=1)          # degenerate: 1 transform per chunk
    assert (a == b).all() and (a == c).all()
    assert (a.reshape(batch, n)[7] == oracle.ntt(x.reshape(batch, n)[7].astype(np.uint64))).all()


def test_host_batch_and_noncanonical_input(ta):
    n, batch = 1 << 9, 6
    ctx = ta.ntt.get_or_create_ctx(n)
    x = oracle.splitmix(n * batch, 17)
    v = x.copy()
    ctx.run_host(v, inverse=False, batch=batch)
    assert (v.reshape(batch, n) == np.stack([oracle.ntt(r) for r in x.reshape(batch, n)])).all()
    # a u64 that is not reduced behaves like BabyBear::new (src/babybear.rs:26-30)
    y = x[:n].copy() + np.uint64(P)
    ta.ntt_gpu(y)
    assert (y == oracle.ntt(x[:n])).all()


def test_u64_device_entry_and_cuda_buffer(ta):
    n = 1 << 10
    x = oracle.splitmix(n, 3)
    buf = ta.CudaBuffer(n)                           # src/ntt.rs:153-215
    buf.copy_from_host(x)
    ctx = ta.ntt.get_or_create_ctx(n)
    ctx.run_device_u64(buf.as_ptr(), 1, False)
    ctx.synchronize()
    out = np.empty(n, dtype=np.uint64)
    buf.copy_to_host(out)
    assert (out == oracle.ntt(x)).all()
    with pytest.raises(AssertionError, match="Size mismatch"):
        buf.copy_from_host(x[:5])
    buf.free()


def test_legacy_void_abi(ta):
    # the reference's own extern block: ntt_ctx_create / ntt_run_inplace / intt_run_inplace (src/ntt.rs:107-109)
    lib = ta._lib.lib
    n = 2048
    ctx = lib.ntt_ctx_create(n)
    assert ctx
    x = oracle.splitmix(n, 8)
    v = x.copy()
    lib.ntt_run_inplace(ctx, v.ctypes.data)
    assert (v == oracle.ntt(x)).all()
    lib.intt_run_inplace(ctx, v.ctypes.data)
    assert (v == x).all()
    lib.ntt_ctx_destroy(ctx)
    lib.ntt_ctx_destroy(None)


def test_context_properties_and_errors(ta):
    assert ta.NttContext(1 << 10).passes == 1
    assert ta.ntt.get_or_create_ctx(1 << 20).passes == 2
    assert ta.ntt.get_or_create_ctx(1 << 21).passes == 3
    assert ta.ntt.get_or_create_ctx(1 << 20) is ta.ntt.get_or_create_ctx(1 << 20)   # per-n cache, src/ntt.rs:128-141
    with pytest.raises(AssertionError, match="power of 2"):
        ta.ntt_gpu(np.zeros(12, dtype=np.uint64))
    with pytest.raises(TypeError):
        ta.ntt_gpu([1, 2, 3, 4])


def test_concurrent_calls_same_n(ta):
    # parallel callers on the same n share one context; results must not interleave
    import threading
    n = 256
    xs = [oracle.splitmix(n, 1000 + t) for t in range(8)]
    outs = [None] * 8

    def work(t):
        for _ in range(20):
            v = xs[t].copy()
            ta.ntt_cuda(v)
            outs[t] = v

    th = [threading.Thread(target=work, args=(t,)) for t in range(8)]
    [t.start() for t in th]
    [t.join() for t in th]
    for t in range(8):
        assert (outs[t] == oracle.ntt(xs[t])).all()


# ---------------------------------------------------------------- coset transforms (BabyBearDomain)
def test_domain_fft_ifft_roundtrip(ta):
    # src/math/domain.rs:193-218
    d = ta.BabyBearDomain(8).with_gpu(True)
    coeffs = [(i * 3 + 1) % P for i in range(8)]
    assert d.ifft(d.fft(coeffs)).tolist() == coeffs
    coset = d.get_coset(7)
    assert coset.ifft(coset.fft(coeffs)).tolist() == coeffs


def test_coset_golden(ta, golden):
    # src/math/domain.rs:220-242: coset FFT == Horner at every coset point
    for c in golden["coset"]:
        d = ta.BabyBearDomain(c["size"]).with_gpu(True).get_coset(c["shift"])
        evals = d.fft(c["coeffs"])
        assert evals.tolist() == c["evals"], c["name"]
        back = d.ifft(evals).tolist()
        assert back[: len(c["coeffs"])] == c["coeffs"] and not any(back[len(c["coeffs"]):])


@pytest.mark.parametrize("log_n", [4, 10, 13, 21])
def test_coset_vs_oracle(ta, log_n):
    n = 1 << log_n
    coeffs = oracle.splitmix(n // 2 + 3 if n > 8 else n, 31)
    d = ta.BabyBearDomain(n).with_gpu(True).get_coset(7)
    evals = d.fft(coeffs)
    assert (evals == oracle.domain_fft(coeffs, n, 7)).all()
    assert (d.ifft(evals) == oracle.domain_ifft(evals, 7)).all()
    got = dev_transform(ta, evals.astype(np.uint32), n, 1, True, shift=7)
    assert (got[: coeffs.size] == coeffs).all() and not got[coeffs.size:].any()


def test_ext_transforms_are_four_base_transforms(ta):
    # src/math/domain.rs:244-278
    d = ta.BabyBearDomain(8).with_gpu(True)
    coeffs = np.array([[(i * 3 + 1) % P, i + 2, i * 7, i + 5] for i in range(8)], dtype=np.uint64)
    evals = d.fft_ext(coeffs)
    for k in range(4):
        assert (evals[:, k] == oracle.domain_fft(coeffs[:, k], 8)).all()
    assert (d.ifft_ext(evals) == coeffs).all()


# ---------------------------------------------------------------- FRI fold
def test_fold_golden(ta, golden):
    for c in golden["fold"]:
        assert ta.fri_fold(c["evals"], c["xs"], c["beta"]).tolist() == c["folded"], c["name"]
        assert ta.fri_fold(c["evals"], c["xs"][: c["n"] // 2], c["beta"]).tolist() == c["folded"]


@pytest.mark.parametrize("m", [2, 6, 10, 4096, 1 << 16])
def test_fold_xs_vs_oracle(ta, m):
    # explicit points need not be a coset: random nonzero xs, any even length
    evals = oracle.splitmix(m, 41)
    xs = oracle.splitmix(m, 42) + np.uint64(1)
    xs %= np.uint64(P)
    xs[xs == 0] = 1
    assert (ta.fri_fold(evals, xs, 987654321) == oracle.fri_fold(evals, xs, 987654321)).all()


def test_fold_errors(ta):
    with pytest.raises(AssertionError, match="even"):               # src/math/fri.rs:28
        ta.fri_fold(np.zeros(5, dtype=np.uint64), np.ones(5, dtype=np.uint64), 1)
    with pytest.raises(AssertionError, match="Cannot invert zero"):  # src/babybear.rs:112
        ta.fri_fold(np.ones(4, dtype=np.uint64), np.array([1, 0], dtype=np.uint64), 1)
    assert ta.fri_fold(np.zeros(0, dtype=np.uint64), np.zeros(0, dtype=np.uint64), 1).size == 0


@pytest.mark.parametrize("log_N,layer", [(1, 0), (3, 0), (10, 0), (10, 3), (10, 9), (16, 2), (21, 0), (21, 5)])
def test_fold_structured_device_vs_oracle(ta, log_N, layer):
    N = 1 << log_N
    m = N >> layer
    ctx = ta.ntt.get_or_create_ctx(N)
    evals = oracle.splitmix(m, 51 + layer)
    x0 = oracle.bb_pow(7, 1 << layer)
    xs = oracle.domain_elements(m, x0)              # x0 * w_m^i
    beta = 555555555
    want = oracle.fri_fold(evals, xs, beta)
    a, o = DevBuf(ta, m * 4), DevBuf(ta, m * 2)
    a.upload(evals.astype(np.uint32))
    ta.fri_fold_device(ctx, a.ptr, o.ptr, m, beta, x0)
    ctx.synchronize()
    got = o.download(np.uint32, m // 2)
    a.free(); o.free()
    assert (got == want).all()


def test_fold_layers_golden_and_constant_final_layer(ta, golden):
    # src/fibonacci.rs:220-245 + src/verifier.rs:69-75
    c = golden["fold_layers"]
    N = c["n"]
    ctx = ta.ntt.get_or_create_ctx(N)
    a, o = DevBuf(ta, N * 4), DevBuf(ta, N * 4)
    a.upload(np.array(c["evals"], dtype=np.uint32))
    ta.fri_fold_layers_device(ctx, a.ptr, o.ptr, c["betas"], c["shift"])
    ctx.synchronize()
    off = 0
    for k, layer in enumerate(c["layers"]):
        got = o.download(np.uint32, len(layer), offset=off * 4)
        assert got.tolist() == layer, f"layer {k}"
        off += len(layer)
    assert len(set(c["layers"][-1])) == 1
    a.free(); o.free()


def test_prover_shape_fold_2_21(ta):
    # BASELINE configs[2] shape: lde 2^21, degree bound 2^17 -> 17 folds 2^21 -> 2^4, final layer constant.
    # Codeword = coset LDE (GPU coset FFT) of a random polynomial of degree < 2^17.
    N, bound, shift = 1 << 21, 1 << 17, 7
    coeffs = oracle.splitmix(bound, 61)
    d = ta.BabyBearDomain(N).with_gpu(True).get_coset(shift)
    evals = d.fft(coeffs)
    betas = (oracle.splitmix(17, 62) % np.uint64(P)).astype(np.uint32)
    ctx = ta.ntt.get_or_create_ctx(N)
    a, o = DevBuf(ta, N * 4), DevBuf(ta, N * 4)
    a.upload(evals.astype(np.uint32))
    ta.fri_fold_layers_device(ctx, a.ptr, o.ptr, betas, shift)
    ctx.synchronize()
    layers = oracle.fri_fold_layers(evals, shift, betas.astype(np.uint64))
    off = 0
    for k, want in enumerate(layers):
        got = o.download(np.uint32, want.size, offset=off * 4)
        assert (got == want).all(), f"layer {k}"
        off += want.size
    final = layers[-1]
    assert final.size == 16 and len(set(final.tolist())) == 1    # src/verifier.rs:69-75
    a.free(); o.free()


# ---------------------------------------------------------------- full sizes through properties
def _spot_check_dft(x32, out32, n, ks):
    """out[k] == sum_j x[j] w^(jk) for a few k, by direct evaluation in numpy (uint64 products < 2^62)."""
    w = oracle.roots_of_unity_domain(n)
    x = x32.astype(np.uint64)
    idx = np.arange(n, dtype=np.uint64)
    for k in ks:
        tw = w[(idx * np.uint64(k)) & np.uint64(n - 1)]
        s = int(((x * tw) % np.uint64(P)).sum(dtype=np.uint64)) % P
        assert int(out32[k]) == s, f"k={k}"


def test_full_size_2_24_properties(ta):
    n = 1 << 24
    x = oracle.splitmix(n, 71).astype(np.uint32)
    y = oracle.splitmix(n, 72).astype(np.uint32)
    fx = dev_transform(ta, x, n, 1, False)
    fy = dev_transform(ta, y, n, 1, False)
    assert (dev_transform(ta, fx, n, 1, True) == x).all()                                   # round trip
    s = ((x.astype(np.uint64) + y) % np.uint64(P)).astype(np.uint32)
    fs = dev_transform(ta, s, n, 1, False)
    assert (fs == ((fx.astype(np.uint64) + fy) % np.uint64(P)).astype(np.uint32)).all()    # linearity
    _spot_check_dft(x, fx, n, [0, 1, 4097, n // 2, n - 1])


def test_full_size_2_27_roundtrip(ta):
    # the field's largest transform (2^28 does not exist)
    n = 1 << 27
    x = oracle.splitmix(n, 81).astype(np.uint32)
    fx = dev_transform(ta, x, n, 1, False)
    assert int(fx[0]) == int(x.sum(dtype=np.uint64)) % P                     # X[0] = sum x
    d1 = np.zeros(n, dtype=np.uint32); d1[1] = 1
    assert (dev_transform(ta, d1, n, 1, False) == oracle.roots_of_unity_domain(n)).all()    # every w^k, k < 2^27
    assert (dev_transform(ta, fx, n, 1, True) == x).all()


@pytest.mark.parametrize("log_n,vecs", [(24, 2), (27, 1)])
def test_full_size_ext_vectors(ta, log_n, vecs):
    """The interleaved (Ext, AoS) passes at the sizes no oracle run can afford: element offsets reach 2^29 words at n = 2^27 (the
    limit of the 32-bit tile-relative byte offsets).  Every coordinate against the single-device BASE transform of that coordinate
    (itself pinned at these sizes by the tests above), direct DFT spot checks, and the round trip."""
    n = 1 << log_n
    rng = np.random.default_rng(2700 + log_n)
    x = rng.integers(0, P, size=(vecs, n, 4), dtype=np.uint32)
    ctx = ta.ntt.get_or_create_ctx(n)
    a, b = DevBuf(ta, x.nbytes), DevBuf(ta, x.nbytes)
    try:
        a.upload(x)
        ctx.run_device_ext_batch(a.ptr, b.ptr, vecs, False)
        ctx.synchronize()
        y = b.download(np.uint32, x.size).reshape(vecs, n, 4)
        ctx.run_device_ext_batch(b.ptr, b.ptr, vecs, True)
        ctx.synchronize()
        assert (b.download(np.uint32, x.size).reshape(x.shape) == x).all(), "round trip"
    finally:
        a.free(); b.free()
    for v in range(vecs):
        for k in range(4):
            col = np.ascontiguousarray(x[v, :, k])
            assert (y[v, :, k] == dev_transform(ta, col, n, 1, False)).all(), (v, k)
    _spot_check_dft(np.ascontiguousarray(x[vecs - 1, :, 3]), np.ascontiguousarray(y[vecs - 1, :, 3]), n, [0, 1, 4097, n // 2, n - 1])


def test_batch_1024_x_2_20(ta):
    # BASELINE configs[3] per-GPU shape: 1024 contiguous transforms of n = 2^20 (4 GiB packed)
    n, batch, block = 1 << 20, 1024, 32
    ctx = ta.ntt.get_or_create_ctx(n)
    xb = oracle.splitmix(n * block, 91).astype(np.uint32)       # 32 distinct transforms, tiled 32x
    buf = DevBuf(ta, n * batch * 4)
    for r in range(batch // block):
        buf.upload(xb, offset=r * xb.nbytes)
    ctx.run_device(buf.ptr, buf.ptr, batch, False)
    ctx.synchronize()
    first = buf.download(np.uint32, n * block)
    for b in (0, 17, 31):
        assert (first[b * n:(b + 1) * n] == oracle.ntt(xb[b * n:(b + 1) * n].astype(np.uint64))).all()
    for r in (1, 13, 31):                                        # every tile position gives the same answer
        assert (buf.download(np.uint32, n * block, offset=r * xb.nbytes) == first).all()
    ctx.run_device(buf.ptr, buf.ptr, batch, True)
    ctx.synchronize()
    for r in (0, 31):
        assert (buf.download(np.uint32, n * block, offset=r * xb.nbytes) == xb).all()
    buf.free()


@pytest.mark.parametrize("log_n,batch", [(20, 3072), (24, 160), (12, 1 << 20), (8, (1 << 24) + 3)])
def test_batches_beyond_4_gib(ta, log_n, batch):
    # sized for 288 GB: 10 - 12 GiB of packed residues in one call, so element indices pass 2^31 and byte offsets 2^32 in
    # every kind of pass (2-pass, 3-pass, the single-sweep kernel, ragged single-pass row tiles); the same 2^24-element
    # block is tiled through the buffer, so every position must reproduce the first block's transform
    n = 1 << log_n
    block_elems = 1 << 24
    per_block = block_elems // n
    xb = oracle.splitmix(block_elems, 95 + log_n).astype(np.uint32)
    total = n * batch
    buf = DevBuf(ta, total * 4)
    off = 0
    while off < total:
        m = min(block_elems, total - off)
        buf.upload(xb[:m], offset=off * 4)
        off += m
    ctx = ta.ntt.get_or_create_ctx(n)
    ctx.run_device(buf.ptr, buf.ptr, batch, False)
    ctx.synchronize()
    first = buf.download(np.uint32, block_elems)
    for b in (0, per_block // 2, per_block - 1):
        assert (first[b * n:(b + 1) * n] == oracle.ntt(xb[b * n:(b + 1) * n].astype(np.uint64))).all()
    nblocks = total // block_elems
    for r in (129, nblocks - 1):                                  # positions beyond 2^31 elements / 2^33 bytes
        assert nblocks > r >= 129 and r * block_elems > (1 << 31)
        assert (buf.download(np.uint32, block_elems, offset=r * block_elems * 4) == first).all()
    tail = total - nblocks * block_elems                          # the ragged end (last case)
    if tail:
        assert (buf.download(np.uint32, tail, offset=nblocks * block_elems * 4) == first[:tail]).all()
    ctx.run_device(buf.ptr, buf.ptr, batch, True)
    ctx.synchronize()
    assert (buf.download(np.uint32, block_elems, offset=(nblocks - 1) * block_elems * 4) == xb).all()
    if tail:
        assert (buf.download(np.uint32, tail, offset=nblocks * block_elems * 4) == xb[:tail]).all()
    buf.free()


# ---------------------------------------------------------------- multi-GPU forms on one GPU (world = 1)
@pytest.mark.parametrize("log_n", [10, 16, 22])
def test_fourstep_world1_matches_oracle(ta, log_n):
    # the 4-step driver of toyni_amd/dist.py with the HIP local stages; the all-to-all degenerates to a copy
    import torch
    from toyni_amd import dist as tdist
    dev = torch.device("cuda", 0)
    n = 1 << log_n
    x = oracle.splitmix(n, 300 + log_n)
    ops = tdist.HipLocalOps(log_n, dev)
    idx_in = tdist.fourstep_input_index(log_n, 1, 0).numpy()
    cols = torch.from_numpy(x[idx_in].astype(np.int32)).to(dev)
    out = tdist.fourstep_forward(cols, log_n, ops)
    torch.cuda.synchronize()
    idx = tdist.fourstep_output_index(log_n, 1, 0).numpy()
    assert (out.cpu().numpy().astype(np.uint64) == oracle.ntt(x)[idx]).all()
    back = tdist.fourstep_inverse(out, log_n, ops)
    torch.cuda.synchronize()
    assert torch.equal(back, cols)


def test_fourstep_2_27_equals_single_device_transform(ta):
    # BASELINE configs[4] at the field's limit n = 2^27, one rank: the 4-step path and the ordinary
    # 3-pass transform must agree element for element
    import torch
    from toyni_amd import dist as tdist
    dev = torch.device("cuda", 0)
    log_n = 27
    n = 1 << log_n
    x32 = oracle.splitmix(n, 2727).astype(np.uint32)
    direct = dev_transform(ta, x32, n, 1, False)
    ops = tdist.HipLocalOps(log_n, dev)
    l1, l2 = tdist.fourstep_split(log_n, 1)
    cols = torch.from_numpy(x32.view(np.int32).reshape(1 << l1, 1 << l2)).to(dev)
    out = tdist.fourstep_forward(cols, log_n, ops)
    torch.cuda.synchronize()
    got = out.cpu().numpy().view(np.uint32)                     # [k1][k2] -> X[k1 + n1 k2]
    assert (got.T.reshape(-1) == direct).all()


# slab form (include/toyni_hip.h 2b): `world` ranks stepped one after the other on the one GPU, the all-to-all done by
# hand -- exercises col_base / row0 / parts of the device stages exactly as a real multi-rank run sets them
@pytest.mark.parametrize("log_n,world", [(13, 1), (14, 4), (16, 2), (20, 1), (20, 8), (21, 4), (24, 8)])
def test_slab_ranks_on_one_gpu_match_oracle(ta, log_n, world):
    import torch
    from toyni_amd import dist as tdist
    dev = torch.device("cuda", 0)
    n = 1 << log_n
    x = oracle.splitmix(n, 700 + log_n)
    want = oracle.ntt(x)
    ops = tdist.HipLocalOps(log_n, dev)
    l1, ls = tdist.slab_split(log_n, world)
    m1, s1 = 1 << l1, 1 << ls
    w, r = s1 // world, m1 // world
    slabs = [torch.from_numpy(x[tdist.slab_input_index(log_n, world, g).numpy()].astype(np.int32)).to(dev) for g in range(world)]
    keep = [t.clone() for t in slabs]
    for g in range(world):
        ops.slab_pass(slabs[g], g * w, False)
    rows = []
    for h in range(world):                                      # rank h receives row block h of every rank's slab
        recv = torch.stack([slabs[g].view(world, r, w)[h] for g in range(world)]).contiguous()
        out = torch.empty((r, s1), dtype=torch.int32, device=dev)
        ops.relayout(recv, out, r, h * r, world, False)
        ops.ntt_rows(out, False)
        rows.append(out)
    torch.cuda.synchronize()
    for h in range(world):
        idx = tdist.slab_output_index(log_n, world, h).numpy()
        assert (rows[h].cpu().numpy().view(np.uint32).astype(np.uint64) == want[idx]).all(), f"forward rank {h}"
    sends = []
    for h in range(world):
        ops.ntt_rows(rows[h], True)
        send = torch.empty((world, r, w), dtype=torch.int32, device=dev)
        ops.relayout(rows[h], send, r, h * r, world, True)
        sends.append(send)
    for g in range(world):
        slab = torch.stack([sends[h][g] for h in range(world)]).reshape(m1, w).contiguous()
        ops.slab_pass(slab, g * w, True)
        torch.cuda.synchronize()
        assert torch.equal(slab, keep[g]), f"inverse rank {g}"


@pytest.mark.parametrize("log_n,chunks", [(16, 1), (20, 4), (22, 8)])
def test_slab_driver_world1_matches_oracle(ta, log_n, chunks):
    # the shipped driver end to end (one rank: the exchange is the identity / local block copies), plain and chunked
    import torch
    from toyni_amd import dist as tdist
    dev = torch.device("cuda", 0)
    n = 1 << log_n
    x = oracle.splitmix(n, 900 + log_n)
    ops = tdist.HipLocalOps(log_n, dev)
    slab = torch.from_numpy(x[tdist.slab_input_index(log_n, 1, 0).numpy()].astype(np.int32)).to(dev)
    keep = slab.clone()
    out = tdist.slab_forward(slab, log_n, ops, chunks=chunks)
    torch.cuda.synchronize()
    idx = tdist.slab_output_index(log_n, 1, 0).numpy()
    assert (out.cpu().numpy().view(np.uint32).astype(np.uint64) == oracle.ntt(x)[idx]).all()
    back = tdist.slab_inverse(out, log_n, ops, chunks=chunks)
    torch.cuda.synchronize()
    assert torch.equal(back, keep)


def test_slab_2_27_equals_single_device_transform(ta):
    # configs[4] at the field's limit, one rank, through the shipped driver (exchange = copy): M1 = 512, S1 = 2^18
    import torch
    from toyni_amd import dist as tdist
    dev = torch.device("cuda", 0)
    log_n = 27
    n = 1 << log_n
    x32 = oracle.splitmix(n, 2728).astype(np.uint32)
    direct = dev_transform(ta, x32, n, 1, False)
    ops = tdist.HipLocalOps(log_n, dev)
    l1, ls = tdist.slab_split(log_n, 1)
    slab = torch.from_numpy(x32.view(np.int32).reshape(1 << l1, 1 << ls)).to(dev)
    out = tdist.slab_forward(slab, log_n, ops)
    torch.cuda.synchronize()
    got = out.cpu().numpy().view(np.uint32)                     # [k1][k'] -> X[k1 + M1 k']
    assert (got.T.reshape(-1) == direct).all()
    back = tdist.slab_inverse(out, log_n, ops)
    torch.cuda.synchronize()
    assert (back.cpu().numpy().view(np.uint32).reshape(-1) == x32).all()


def test_slab_entry_points_reject_bad_shapes(ta):
    from toyni_amd._lib import lib
    small = ta.ntt.get_or_create_ctx(1024)
    assert lib.toyni_ntt_ctx_first_pass_points(small.handle) == 0
    big = ta.ntt.get_or_create_ctx(1 << 16)
    assert lib.toyni_ntt_ctx_first_pass_points(big.handle) == 256
    buf = ta.GpuBuffer(1 << 16)
    assert lib.toyni_ntt_slab_pass_device(small.handle, buf.ptr, 32, 0, 0, None) == 10001      # single-pass plan
    assert lib.toyni_ntt_slab_pass_device(big.handle, buf.ptr, 16, 0, 0, None) == 10006        # < 32 columns
    assert lib.toyni_ntt_slab_pass_device(big.handle, buf.ptr, 48, 0, 0, None) == 10006        # not a power of two
    assert lib.toyni_ntt_slab_pass_device(big.handle, buf.ptr, 128, 192, 0, None) == 10006     # runs past column S1
    assert lib.toyni_ntt_slab_relayout_device(big.handle, buf.ptr, buf.ptr, 256, 0, 1, 0, None) == 10006   # in place
    buf.free()


def _run_on_measurement_build(code, extra_env=None, timeout=300):
    """Runs `code` in a child interpreter whose toyni_amd is bound to libtoyni_hip_tools.so (the measurement build of the same
    source, include/toyni_hip_tools.h): the launch-timing hooks live there, not in the shipped library."""
    import os
    import subprocess
    import sys
    import __graft_entry__ as entry
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pp = os.pathsep.join([root, os.path.join(root, "tests"), os.path.join(root, "tests", "_hooks")])   # _hooks: the child reports its launches
    env = dict(os.environ, PYTHONPATH=pp, TOYNI_LIB_OVERRIDE=entry.build_tools(), **(extra_env or {}))
    return subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=timeout, env=env, cwd=root)


def test_shipped_library_has_no_measurement_hooks(ta):
    assert not ta._lib.HAS_TOOLS and not hasattr(ta._lib.lib, "toyni_ntt_profile_passes")
    with pytest.raises(RuntimeError, match="measurement build"):
        ta.ntt.get_or_create_ctx(1 << 12).timing(True)


def test_both_executors_of_the_single_sweep_sizes(ta):
    # n = 2^13 .. 2^15 have two executors: the two-pass plan and the single-sweep LDS kernel (by default used for large
    # batches of 2^13 only; n = 2^11 / 2^12 have the one- / two-waves-per-transform kernels, tests/test_gpu_stream3.py).  Force each one for every size, small ragged batches, in child processes (the knobs are
    # read once per process); which executor ran is read off the launch records (1 launch per transform vs 2).
    code = (
        "import os, numpy as np, oracle, toyni_amd\n"
        "from test_gpu_parity import DevBuf\n"
        "want_launches = int(os.environ['TOYNI_TEST_LAUNCHES'])\n"
        "for log_n in range(13 if want_launches == 1 else 11, 16):   # (n = 2^11 / 2^12 have no LDS-kernel form since round 5: Row2048 / Row4096)\n"
        "    n, batch = 1 << log_n, 5\n"
        "    ctx = toyni_amd.NttContext(n)\n"
        "    x = oracle.splitmix(batch * n, 777 + log_n).astype(np.uint32)\n"
        "    a, o = DevBuf(toyni_amd, x.nbytes), DevBuf(toyni_amd, x.nbytes)\n"
        "    a.upload(x)\n"
        "    ctx.timing(True)\n"
        "    ctx.run_device(a.ptr, o.ptr, batch, False)\n"
        "    t = ctx.read_timing()\n"
        "    assert sum(t['launches']['forward']) == want_launches, (log_n, t['launches'])\n"
        "    f = o.download(np.uint32, x.size)\n"
        "    ctx.run_device(a.ptr, a.ptr, batch, True, shift=7)\n"
        "    ctx.synchronize()\n"
        "    i = a.download(np.uint32, x.size)\n"
        "    for b in range(batch):\n"
        "        row = x[b * n:(b + 1) * n].astype(np.uint64)\n"
        "        assert (f[b * n:(b + 1) * n] == oracle.ntt(row)).all()\n"
        "        assert (i[b * n:(b + 1) * n] == oracle.domain_ifft(row, 7)).all()\n"
        "print('EXECUTOR OK')\n")
    modes = [("two-pass", {"TOYNI_NO_LDS_KERNEL": "1", "TOYNI_TEST_LAUNCHES": "2"}),
             # the non-temporal kernel variants (normally chosen for footprints >= 512 MiB) forced on small data
             ("two-pass, non-temporal", {"TOYNI_NO_LDS_KERNEL": "1", "TOYNI_NT_MIN_BYTES": "0", "TOYNI_TEST_LAUNCHES": "2"}),
             ("sweep, 4 workgroups per CU", {"TOYNI_LDS_MAX_LOG": "15", "TOYNI_LDS_MIN_ELEMS": "0", "TOYNI_TEST_LAUNCHES": "1"}),
             ("sweep, 1 workgroup per CU", {"TOYNI_LDS_MAX_LOG": "15", "TOYNI_LDS_MIN_ELEMS": "0", "TOYNI_LDS_ROWS": "5", "TOYNI_TEST_LAUNCHES": "1"})]
    for mode, extra in modes:
        res = _run_on_measurement_build(code, extra)
        assert res.returncode == 0 and "EXECUTOR OK" in res.stdout, mode + ": " + res.stdout[-1000:] + res.stderr[-3000:]


def test_large_batches_of_mid_sizes_take_the_single_sweep_kernel(ta):
    # the default policy: >= 2^25 elements of n = 2^13 per call -> one launch of the single-sweep LDS kernel (counted on the
    # measurement build); checked against the oracle on sampled rows
    code = (
        "import numpy as np, oracle, toyni_amd\n"
        "from test_gpu_parity import DevBuf\n"
        "n, batch = 1 << 13, 1 << 12\n"
        "x = np.random.default_rng(12).integers(0, oracle.P, size=n * batch, dtype=np.uint32)\n"
        "ctx = toyni_amd.NttContext(n)\n"
        "buf = DevBuf(toyni_amd, x.nbytes)\n"
        "buf.upload(x)\n"
        "ctx.timing(True)\n"
        "ctx.run_device(buf.ptr, buf.ptr, batch, False)\n"
        "assert ctx.read_timing()['launches']['forward'] == [1, 0]\n"
        "ctx.timing(False)\n"
        "got = buf.download(np.uint32, x.size)\n"
        "for b in (0, 1, 7, 2047, 2048, batch - 1):\n"
        "    assert (got[b * n:(b + 1) * n] == oracle.ntt(x[b * n:(b + 1) * n].astype(np.uint64))).all(), b\n"
        "ctx.run_device(buf.ptr, buf.ptr, batch, True)\n"
        "ctx.synchronize()\n"
        "assert (buf.download(np.uint32, x.size) == x).all()\n"
        "print('SWEEP OK')\n")
    res = _run_on_measurement_build(code)
    assert res.returncode == 0 and "SWEEP OK" in res.stdout, res.stdout[-1000:] + res.stderr[-3000:]


# ---------------------------------------------------------------- low-degree extension (src/fibonacci.rs:101-103)
@pytest.mark.parametrize("log_n,log_blowup,batch,shift", [
    (11, 5, 3, 7), (12, 1, 2, 7), (14, 3, 5, 1), (16, 5, 4, 7), (16, 8, 2, 7), (18, 2, 3, 7), (20, 5, 2, 7), (20, 1, 1, 7),
    (20, 4, 16, 7), (21, 5, 2, 7), (21, 7, 1, 1234567), (22, 3, 1, 7),
    (8, 3, 5, 7), (10, 5, 2, 7), (3, 3, 2, 7), (16, 12, 2, 7), (20, 0, 1, 7)])   # last row: pad-and-transform fallbacks, blowup 1
def test_lde_matches_padded_transform_and_oracle(ta, log_n, log_blowup, batch, shift):
    n, n_in = 1 << log_n, (1 << log_n) >> log_blowup
    coeffs = oracle.splitmix(n_in * batch, 5000 + 64 * log_n + log_blowup).astype(np.uint32)
    ctx = ta.ntt.get_or_create_ctx(n)
    a, o = DevBuf(ta, coeffs.nbytes), DevBuf(ta, 4 * n * batch)
    try:
        a.upload(coeffs)
        o.upload(np.full(n * batch, 0xDEADBEEF, dtype=np.uint32))      # every output word must be written
        ctx.lde_device(a.ptr, o.ptr, batch, log_blowup, shift)
        ctx.synchronize()
        got = o.download(np.uint32, n * batch)
        assert (a.download(np.uint32, coeffs.size) == coeffs).all()
    finally:
        a.free()
        o.free()
    for b in range(batch):
        want = oracle.domain_fft(coeffs[b * n_in:(b + 1) * n_in].astype(np.uint64), n, shift)
        assert (got[b * n:(b + 1) * n].astype(np.uint64) == want).all(), f"transform {b}"
    padded = np.zeros((batch, n), dtype=np.uint32)
    padded[:, :n_in] = coeffs.reshape(batch, n_in)
    assert (dev_transform(ta, padded.reshape(-1), n, batch, False, shift=shift) == got).all()


@pytest.mark.parametrize("log_n,ncoeffs", [(0, 1), (3, 3), (8, 0), (8, 256), (10, 40), (11, 1), (12, 100), (16, 2048), (16, 2049),
                                           (18, 8192), (20, 32768), (20, 1 << 20), (21, 65676)])
def test_lde_host_is_domain_fft(ta, log_n, ncoeffs):
    # BabyBearDomain::fft(coeffs), src/math/domain.rs:107-123, through the one-call host entry point: any coefficient
    # count (the device pads the compact vector up to a power of two, the rest of the padding is implied)
    n = 1 << log_n
    coeffs = oracle.splitmix(ncoeffs, 6000 + log_n) if ncoeffs else np.zeros(0, dtype=np.uint64)
    got = ta.ntt.get_or_create_ctx(n).lde_host(coeffs, shift=7)
    want = oracle.domain_fft(coeffs, n, 7) if ncoeffs else np.zeros(n, dtype=np.uint64)
    assert got.dtype == np.uint64 and (got == want).all()
    if ncoeffs:                                      # non-canonical u64 inputs behave like BabyBear::new (src/babybear.rs:26-30)
        big = coeffs + np.uint64(P) * np.uint64(3)
        assert (ta.ntt.get_or_create_ctx(n).lde_host(big, shift=7) == want).all()
        reuse = np.full(n, 0xdead, dtype=np.uint64)  # the caller's own output array, reused
        assert ta.ntt.get_or_create_ctx(n).lde_host(coeffs, shift=7, out=reuse) is reuse and (reuse == want).all()
        if n >= 2:
            xs = oracle.domain_elements(n, 7)[: n // 2]
            half = np.empty(n // 2, dtype=np.uint64)
            assert ta.fri_fold(want, xs, 99, out=half) is half and (half == oracle.fri_fold(want, xs, 99)).all()


@pytest.mark.parametrize("log_n,ncoeffs", [(0, 1), (4, 5), (10, 0), (10, 1000), (12, 64), (16, 3000), (20, 1 << 15), (21, 1 << 21)])
def test_lde_ext_is_four_base_extensions(ta, log_n, ncoeffs):
    # fft_ext (src/math/domain.rs:134-151) = the four coordinate columns through fft; host and device forms
    n = 1 << log_n
    c4 = oracle.splitmix(4 * ncoeffs, 6500 + log_n).reshape(ncoeffs, 4) if ncoeffs else np.zeros((0, 4), dtype=np.uint64)
    ctx = ta.ntt.get_or_create_ctx(n)
    got = ctx.lde_ext_host(c4, shift=7)
    assert got.shape == (n, 4)
    for k in range(4):
        want = oracle.domain_fft(np.ascontiguousarray(c4[:, k]), n, 7) if ncoeffs else np.zeros(n, dtype=np.uint64)
        assert (got[:, k] == want).all(), f"coordinate {k}"
    if ncoeffs and ncoeffs & (ncoeffs - 1) == 0:             # device form: power-of-two coefficient counts
        a, o = DevBuf(ta, 16 * ncoeffs), DevBuf(ta, 16 * n)
        a.upload(c4.astype(np.uint32))
        check = ta._lib.check
        check(ta._lib.lib.toyni_lde_ext_device(ctx.handle, a.ptr, o.ptr, log_n - (ncoeffs.bit_length() - 1), 7, None), "lde ext")
        ctx.synchronize()
        dev = o.download(np.uint32, 4 * n).reshape(n, 4)
        a.free()
        o.free()
        assert (dev.astype(np.uint64) == got).all()
    # and the domain mirror (what a caller of BabyBearDomain::fft_ext sees)
    assert (ta.BabyBearDomain(n).with_gpu(True).get_coset(7).fft_ext(c4) == got).all()


def test_lde_rejects_bad_arguments(ta):
    ctx = ta.ntt.get_or_create_ctx(1 << 12)
    buf = DevBuf(ta, 4 << 12)
    lib = ta._lib.lib
    assert lib.toyni_lde_device(ctx.handle, buf.ptr, buf.ptr, 1, 2, 7, None) == 10006       # in place with a blow-up
    assert lib.toyni_lde_device(ctx.handle, buf.ptr, buf.ptr, 1, 13, 7, None) == 10006      # blow-up larger than n
    assert lib.toyni_lde_device(ctx.handle, buf.ptr, buf.ptr, 1, 1, 0, None) == 10006       # shift 0
    assert lib.toyni_lde_device(ctx.handle, None, buf.ptr, 1, 1, 7, None) == 10002
    host = np.zeros(1 << 13, dtype=np.uint64)
    assert lib.toyni_lde_host(ctx.handle, host.ctypes.data, (1 << 12) + 1, host.ctypes.data, 7) == 10006   # more coefficients than points
    assert lib.toyni_lde_host(ctx.handle, host.ctypes.data, 4, host.ctypes.data, 0) == 10006                # shift 0
    buf.free()


@pytest.mark.parametrize("log_ctx,log_m,shift", [(0, 0, 1), (4, 4, 1), (10, 3, 7), (16, 16, 7), (21, 21, 7), (21, 12, 1234567), (27, 20, 7)])
def test_domain_elements_match_reference_chain(ta, log_ctx, log_m, shift):
    # roots_of_unity_domain (src/ntt.rs:69-81) / BabyBearDomain::elements (src/math/domain.rs:61-69)
    ctx = ta.ntt.get_or_create_ctx(1 << log_ctx)
    m = 1 << log_m
    buf = DevBuf(ta, 4 * m)
    ctx.domain_elements_device(buf.ptr, m, shift)
    ctx.synchronize()
    got = buf.download(np.uint32, m)
    buf.free()
    assert (got.astype(np.uint64) == oracle.domain_elements(m, shift)).all()
    if shift == 1:
        assert (got.astype(np.uint64) == oracle.roots_of_unity_domain(m)).all()


def test_in_workload_launch_timing(ta):
    # measurement build, toyni_ntt_ctx_timing: every pass launch between enable and read is bracketed by events; results are
    # untouched; a lone 2^16 transform takes the three-step latency shapes (still two launches)
    code = (
        "import numpy as np, oracle, toyni_amd\n"
        "from test_gpu_parity import DevBuf\n"
        "n, batch = 1 << 16, 8\n"
        "x = oracle.splitmix(n * batch, 4242).astype(np.uint32)\n"
        "ctx = toyni_amd.NttContext(n)\n"
        "buf = DevBuf(toyni_amd, x.nbytes)\n"
        "buf.upload(x)\n"
        "ctx.timing(True)\n"
        "ctx.run_device(buf.ptr, buf.ptr, batch, False)\n"
        "ctx.run_device(buf.ptr, buf.ptr, batch, True)\n"
        "ctx.run_device(buf.ptr, buf.ptr, batch, False)\n"
        "t = ctx.read_timing()\n"
        "ctx.timing(False)\n"
        "assert t['launches'] == {'forward': [2, 2], 'inverse': [1, 1]}, t\n"
        "assert all(v is not None and 0.0 < v < 50.0 for v in t['forward'] + t['inverse'])\n"
        "ctx.run_device(buf.ptr, buf.ptr, batch, True)\n"
        "ctx.synchronize()\n"
        "assert ctx.read_timing()['launches'] == {'forward': [0, 0], 'inverse': [0, 0]}\n"
        "assert (buf.download(np.uint32, n * batch) == x).all()\n"
        "p = ctx.profile_passes(buf.ptr, batch, False, reps=3)\n"
        "assert len(p) == 2 and all(0.0 < v < 50.0 for v in p)\n"
        "print('TIMING OK')\n")
    res = _run_on_measurement_build(code)
    assert res.returncode == 0 and "TIMING OK" in res.stdout, res.stdout[-1000:] + res.stderr[-3000:]


# ---------------------------------------------------------------- Ext-valued fold (src/math/fri.rs:7-25)
@pytest.mark.parametrize("m", [2, 8, 1024, 1 << 15])
def test_fold_ext_vs_oracle(ta, m):
    rng = np.random.default_rng(m)
    evals = rng.integers(0, P, size=(m, 4)).astype(np.uint64)
    xs = oracle.domain_elements(m, 7)
    beta = [int(v) for v in rng.integers(0, P, size=4)]
    want = oracle.fri_fold_ext(evals, xs, beta)
    assert (ta.fri_fold_ext(evals, xs, beta) == want).all()                       # explicit points, host form
    # structured points, device-resident
    ctx = ta.ntt.get_or_create_ctx(max(m, 2))
    a, o = DevBuf(ta, m * 16), DevBuf(ta, m * 8)
    a.upload(evals.astype(np.uint32))
    ta.fri_fold_ext_device(ctx, a.ptr, o.ptr, m, beta, 7)
    ctx.synchronize()
    got = o.download(np.uint32, (m // 2) * 4).reshape(-1, 4)
    a.free(); o.free()
    assert (got == want).all()


def test_fold_ext_embeds_base_fold(ta):
    # a base codeword embedded in Ext with a base beta folds to the embedded base fold
    m = 256
    evals = oracle.splitmix(m, 9)
    xs = oracle.domain_elements(m, 7)
    e4 = np.zeros((m, 4), dtype=np.uint64)
    e4[:, 0] = evals
    out = ta.fri_fold_ext(e4, xs, [424242, 0, 0, 0])
    assert (out[:, 0] == oracle.fri_fold(evals, xs, 424242)).all() and not out[:, 1:].any()
    with pytest.raises(AssertionError, match="even"):
        ta.fri_fold_ext(np.zeros((3, 4), dtype=np.uint64), np.ones(3, dtype=np.uint64), [1, 0, 0, 0])


@pytest.mark.parametrize("log_n,shift", [(0, 1), (1, 7), (3, 1), (5, 7), (7, 1), (10, 7), (11, 1), (13, 7), (16, 7), (19, 3), (20, 7), (21, 1)])
def test_ext_transform_single_call(ta, log_n, shift):
    # fft_ext / ifft_ext (src/math/domain.rs:129-151) as one device call: equals four base transforms of the coordinates
    n = 1 << log_n
    rng = np.random.default_rng(log_n)
    vals = rng.integers(0, P, size=(n, 4)).astype(np.uint64)
    d = ta.BabyBearDomain(n).with_gpu(True).get_coset(shift)
    evals = d.fft_ext(vals)
    for k in range(4):
        assert (evals[:, k] == oracle.domain_fft(vals[:, k], n, shift)).all()
    assert (d.ifft_ext(evals) == vals).all()
    # device-resident AoS form
    ctx = ta.ntt.get_or_create_ctx(n)
    buf = DevBuf(ta, n * 16)
    buf.upload(vals.astype(np.uint32))
    ctx.run_device_ext(buf.ptr, False, shift=shift)
    ctx.synchronize()
    assert (buf.download(np.uint32, 4 * n).reshape(n, 4) == evals).all()
    buf.free()


@pytest.mark.parametrize("log_n,batch,shift", [(2, 9, 1), (6, 33, 7), (10, 5, 7), (10, 64, 1), (12, 7, 7), (14, 40, 1), (16, 9, 7), (18, 16, 5), (20, 3, 7), (20, 16, 1)])
def test_ext_batched_device_resident_vs_oracle(ta, log_n, batch, shift):
    # `batch` Ext vectors ([batch][n][4] packed u32) through toyni_ntt_ext_batch_device: every coordinate of every vector is the
    # oracle's coset transform of that coordinate column; out of place forward, in place inverse
    n = 1 << log_n
    rng = np.random.default_rng(1000 * log_n + batch)
    x = rng.integers(0, P, size=(batch, n, 4), dtype=np.uint32)
    ctx = ta.ntt.get_or_create_ctx(n)
    a, b = DevBuf(ta, x.nbytes), DevBuf(ta, x.nbytes)
    try:
        a.upload(x)
        ctx.run_device_ext_batch(a.ptr, b.ptr, batch, False, shift=shift)
        ctx.synchronize()
        y = b.download(np.uint32, x.size).reshape(batch, n, 4)
        assert (a.download(np.uint32, x.size).reshape(x.shape) == x).all(), "out-of-place transform modified its input"
        for v in range(batch):
            for k in range(4):
                assert (y[v, :, k] == oracle.domain_fft(x[v, :, k].astype(np.uint64), n, shift)).all(), (v, k)
        ctx.run_device_ext_batch(b.ptr, b.ptr, batch, True, shift=shift)
        ctx.synchronize()
        assert (b.download(np.uint32, x.size).reshape(x.shape) == x).all()
    finally:
        a.free(); b.free()


def test_ext_transform_launches_exactly_the_plans_passes(ta):
    # no de-interleave sweep on either side -- an Ext transform is the plan's passes and nothing else
    code = """
import numpy as np, ctypes
import toyni_amd
from toyni_amd import _lib
from test_gpu_parity import DevBuf
for log_n, batch in ((10, 3), (16, 1), (20, 2), (21, 1)):
    n = 1 << log_n
    ctx = toyni_amd.ntt.get_or_create_ctx(n)
    buf = DevBuf(toyni_amd, 16 * n * batch)
    buf.upload(np.arange(4 * n * batch, dtype=np.uint32) % 2013265921)
    ctx.run_device_ext_batch(buf.ptr, buf.ptr, batch, False, shift=7)   # warm: tables, scratch
    ctx.synchronize()
    before = set(_lib.launched_kernels())
    ctx.timing(True)
    ctx.run_device_ext_batch(buf.ptr, buf.ptr, batch, False, shift=7)
    ctx.run_device_ext_batch(buf.ptr, buf.ptr, batch, True, shift=7)
    t = ctx.read_timing()
    ctx.timing(False)
    want = ctx.passes_for(4 * batch)      # n = 2^21: the two-pass plan (1024-point column pass + streaming 2048-point closing pass), Ext vectors too
    assert want == (2 if log_n == 21 else ctx.passes)
    for direction in ("forward", "inverse"):
        assert t["launches"][direction] == [1] * want + [0] * (ctx.passes - want), (log_n, t)
    assert set(_lib.launched_kernels()) == before, "an Ext transform launched something besides its passes"
    buf.free()
print("EXT LAUNCHES OK")
"""
    res = _run_on_measurement_build(code)
    assert res.returncode == 0 and "EXT LAUNCHES OK" in res.stdout, res.stdout[-2000:] + res.stderr[-3000:]


def test_single_process_multi_gpu_batch_runner(ta):
    # toyni_ntt_host_multi_gpu: contiguous shards, one host thread + context per listed device.  One GPU here, so the
    # device is listed three times: the sharding (ragged: 7 = 3 + 2 + 2), the threads and the per-device contexts all run.
    n, batch = 1 << 12, 7
    x = oracle.splitmix(n * batch, 606)
    v = x.copy()
    devs = (ctypes.c_int * 3)(0, 0, 0)
    rc = ta._lib.lib.toyni_ntt_host_multi_gpu(devs, 3, n, v.ctypes.data, batch, 0)
    assert rc == 0, ta._lib.error_string(rc)
    assert (v.reshape(batch, n) == np.stack([oracle.ntt(r) for r in x.reshape(batch, n)])).all()
    rc = ta._lib.lib.toyni_ntt_host_multi_gpu(devs, 3, n, v.ctypes.data, batch, 1)
    assert rc == 0 and (v == x).all()
    bad = (ctypes.c_int * 1)(99)
    assert ta._lib.lib.toyni_ntt_host_multi_gpu(bad, 1, n, v.ctypes.data, 1, 0) == 10006   # TOYNI_E_RANGE: no such device


def test_fold_ext_and_merkle_golden(ta, golden):
    for c in golden["ext"]["fold"]:
        assert ta.fri_fold_ext(np.array(c["evals"], dtype=np.uint64), c["xs"], c["beta"]).tolist() == c["folded"], c["name"]
    for c in golden["merkle"]:
        salts = np.frombuffer(b"".join(bytes.fromhex(s) for s in c["salts_hex"]), dtype=np.uint8).reshape(c["n"], 16)
        assert ta.MerkleTree(c["values"]).root().hex() == c["root_unsalted"]
        assert ta.MerkleTree(c["values"], salts).root().hex() == c["root_salted"]


@pytest.mark.parametrize("log_n", [2, 3, 4, 5, 6, 10, 12])
@pytest.mark.parametrize("misalign_words", [1, 2, 3])
def test_device_buffers_need_only_word_alignment(ta, log_n, misalign_words):
    """The packed-u32 device entry points take any 4-byte aligned pointer: the single-step shapes read and write a thread's row
    with 16-byte accesses, which must not assume more alignment than that.  Ragged batch, out of place and in place, coset inverse."""
    n, batch = 1 << log_n, 37
    x = oracle.splitmix(n * batch, 1300 + log_n).astype(np.uint32)
    ctx = ta.ntt.get_or_create_ctx(n)
    pad = 4 * misalign_words
    a, o = DevBuf(ta, x.nbytes + 16), DevBuf(ta, x.nbytes + 16)
    try:
        a.upload(x, offset=pad)
        ctx.run_device(a.ptr + pad, o.ptr + pad, batch, False)
        ctx.synchronize()
        got = o.download(np.uint32, x.size, offset=pad)
        for b in (0, 1, batch - 1):
            assert (got[b * n:(b + 1) * n] == oracle.ntt(x[b * n:(b + 1) * n].astype(np.uint64))).all(), b
        ctx.run_device(o.ptr + pad, o.ptr + pad, batch, True)                # inverse in place: back to the input
        ctx.synchronize()
        assert (o.download(np.uint32, x.size, offset=pad) == x).all()
        ctx.run_device(a.ptr + pad, a.ptr + pad, batch, True, shift=7)       # coset inverse: the scaled / twiddled store path
        ctx.synchronize()
        got = a.download(np.uint32, x.size, offset=pad)
        for b in (0, batch - 1):
            assert (got[b * n:(b + 1) * n] == oracle.domain_ifft(x[b * n:(b + 1) * n].astype(np.uint64), 7)).all(), b
    finally:
        a.free()
        o.free()
